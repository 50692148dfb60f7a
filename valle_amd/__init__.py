"""MI355X-native VALL-E decode engine (AR + NAR hot path of lifeiteng/vall-e).

    from valle_amd import get_model, VALLE, Engine, EngineConfig, ops, modules

Everything numeric runs in ``libvalle_engine.so`` (hand-written HIP for gfx950); this package is
the host-side mirror of the reference's Python interface for that path.
"""
from . import _lib, codec, modules, ops  # noqa: F401
from .codec import EncodecDecoder  # noqa: F401
from .engine import Engine, EngineConfig, sine_pe  # noqa: F401
from .model import VALLE, VALLF, PromptedFeatures, add_model_arguments, get_model  # noqa: F401
from .serving import ContinuousBatcher, Request  # noqa: F401
from .formats import TextTokenCollater, get_text_token_collater, load_checkpoint, read_symbol_table, save_checkpoint  # noqa: F401

__all__ = ["VALLE", "VALLF", "PromptedFeatures", "get_model", "add_model_arguments", "load_checkpoint", "save_checkpoint", "TextTokenCollater", "ContinuousBatcher", "Request",
           "get_text_token_collater", "read_symbol_table", "Engine", "EngineConfig", "ops", "modules", "sine_pe", "codec", "EncodecDecoder"]
