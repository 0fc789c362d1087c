"""Reference module path `utils.load_pretrained` (src/utils/load_pretrained.py:8-156) -> gaviko_amd.utils.load_pretrained (offline loader)."""
from gaviko_amd.utils.load_pretrained import *  # noqa: F401,F403
from gaviko_amd.utils.load_pretrained import (load_pretrain, load_vanilla_pretrain, load_vanilla_pretrain_with_adapters,  # noqa: F401
                                              mapping_vit)
