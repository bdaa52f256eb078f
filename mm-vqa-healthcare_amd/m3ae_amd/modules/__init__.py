"""Mirror of the reference's m3ae.modules package surface for the hot path (m3ae/modules/__init__.py:1-5)."""
from .m3ae_module import M3AETransformerSS, state_dict_spec  # noqa: F401
from .m3ae_t5_mm_encoder_input import T5VQA_MMEncoderInput  # noqa: F401
from .m3ae_decoder import DecoderModel  # noqa: F401
