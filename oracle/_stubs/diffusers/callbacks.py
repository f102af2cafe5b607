"""names the reference's SD3 pipeline imports (custom_pipeline.py:8); never instantiated on the UniVST path"""


class PipelineCallback:
    tensor_inputs = []


class MultiPipelineCallbacks:
    tensor_inputs = []
