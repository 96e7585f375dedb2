"""Wire formats of real checkpoints for the decode hot path: per-layer fp8 KV-cache scales
(kv_cache_scales.json) and the AWQ / GPTQ int4 tensor layouts with their tensor-parallel
sharding rules."""
from .kv_scales import kv_cache_scales_loader, load_kv_cache_scales  # noqa: F401
from .quant_checkpoint import (QuantParamSpec, load_llama_layer_weights,  # noqa: F401
                               param_specs, shard_column, shard_merged_column, shard_qkv,
                               shard_row)
