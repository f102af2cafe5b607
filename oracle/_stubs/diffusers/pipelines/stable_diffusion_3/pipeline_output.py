from dataclasses import dataclass


@dataclass
class StableDiffusion3PipelineOutput:
    images: object
