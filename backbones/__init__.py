"""drop-in shim: the reference import path re-exporting the native implementation in univst_amd."""
