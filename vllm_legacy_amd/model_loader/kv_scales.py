"""Reader of kv_cache_scales.json, the per-layer fp8 KV-cache scaling factors of a checkpoint.

Same contract as the reference's kv_cache_scales_loader
(vllm/model_executor/model_loader/weight_utils.py:365-402) and its schema
(vllm/model_executor/layers/quantization/schema.py:19-84):

    {"model_type": "llama",
     "kv_cache": {"dtype": "float8_e4m3fn",
                  "scaling_factor": {"<tp_rank>": {"<layer>": <float>, ...}, ...}}}

Validation failures do not raise: they are logged and an empty list comes back, which leaves
every layer at scale 1.0 -- the reference's behaviour.

Convention: quantised_value * scaling_factor ~= true_value.  The reference doubles the factor on
ROCm because its FNUZ cache format tops out at 240 instead of 448
(vllm/model_executor/models/llama.py:483-488); this backend stores OCP e4m3fn, the format the
file declares, so the factor is used as written.
"""
import json
import logging
from typing import Iterable, List, Optional, Tuple

logger = logging.getLogger(__name__)


class KVScalesError(ValueError):
    pass


def _validate(doc: dict, tp_rank: int, tp_size: int, num_hidden_layers: int,
              model_type: Optional[str]) -> List[Tuple[int, float]]:
    if not isinstance(doc, dict) or "kv_cache" not in doc or "model_type" not in doc:
        raise KVScalesError("missing 'model_type' or 'kv_cache'")
    if model_type is not None and doc["model_type"] != model_type:
        raise KVScalesError(f"Model type is {model_type} but loaded scaling factors belonging to "
                            f"different model type {doc['model_type']}!")
    kvc = doc["kv_cache"]
    if kvc.get("dtype") != "float8_e4m3fn":
        raise KVScalesError("Loaded scaling factors intended for KV cache dtype = "
                            f"{kvc.get('dtype')} rather than float8_e4m3fn!")
    try:
        table = {int(r): {int(l): float(s) for l, s in layers.items()}
                 for r, layers in kvc["scaling_factor"].items()}
    except (KeyError, TypeError, ValueError, AttributeError) as e:
        raise KVScalesError(f"malformed scaling_factor table: {e}") from None
    if len(table) != tp_size:
        raise KVScalesError(f"Loaded dictionary has TP size {len(table)} but LLM engine is "
                            f"currently running with TP size {tp_size}.")
    for r, layers in table.items():
        if len(layers) != num_hidden_layers:
            raise KVScalesError(f"KV cache scales map for TP rank {r} is malformed. Expected "
                                f"{num_hidden_layers} layers, got {len(layers)}.")
    for r in range(tp_size):
        if r not in table:
            raise KVScalesError(f"KV cache scales map for TP rank {r} not found.")
    mine = table[tp_rank]
    for i in range(num_hidden_layers):
        if i not in mine:
            raise KVScalesError(f"Could not find KV cache scales for layer {i} in TP rank {tp_rank}.")
    return list(mine.items())


def kv_cache_scales_loader(filename: str, tp_rank: int, tp_size: int, num_hidden_layers: int,
                           model_type: Optional[str]) -> Iterable[Tuple[int, float]]:
    try:
        with open(filename) as f:
            doc = json.load(f)
        return _validate(doc, tp_rank, tp_size, num_hidden_layers, model_type)
    except FileNotFoundError:
        logger.error("File or directory '%s' not found.", filename)
    except json.JSONDecodeError:
        logger.error("Error decoding JSON in file '%s'.", filename)
    except Exception as e:  # the reference swallows every failure here
        logger.error("An error occurred while reading '%s': %s", filename, e)
    logger.warning("Defaulting to KV cache scaling factors = 1.0 for all layers in TP rank %d as "
                   "an error occurred during loading.", tp_rank)
    return []


def load_kv_cache_scales(layers, quantization_param_path: str, tp_rank: int, tp_size: int,
                         model_type: Optional[str] = "llama") -> int:
    """Set `layer.attn._kv_scale` of every decoder layer (models/llama.py:473-493).  Returns the
    number of layers that received a factor (0 when the file was rejected)."""
    n = 0
    for layer_idx, scaling_factor in kv_cache_scales_loader(quantization_param_path, tp_rank,
                                                            tp_size, len(layers), model_type):
        attn = getattr(layers[layer_idx], "attn", None)
        if attn is None or not hasattr(attn, "_kv_scale"):
            raise RuntimeError("Self attention has no KV cache scaling factor attribute!")
        attn._kv_scale = float(scaling_factor)
        n += 1
    return n
