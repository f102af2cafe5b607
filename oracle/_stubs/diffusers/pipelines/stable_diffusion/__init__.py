from dataclasses import dataclass
from typing import Any


@dataclass
class StableDiffusionPipelineOutput:
    images: Any
    nsfw_content_detected: Any = None
