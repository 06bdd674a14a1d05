"""evogp_amd.pipeline — the generation loop (reference: src/evogp/pipeline/)."""
from .standard import BasePipeline, StandardPipeline

__all__ = ["BasePipeline", "StandardPipeline"]
