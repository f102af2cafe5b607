def is_compiled_module(m):
    return False
