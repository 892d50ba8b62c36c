"""keds_amd -- MI355X-native retrieval hot path of KEDs (suoych/KEDs) behind the reference's model API.

    from keds_amd import CLIP, IM2TEXT, CrossFormer, build_model, FlatIndex, IndexFlatL2

Python here is the host side only (parameter containers, argument checks, torch device memory and
torch.distributed); all compute runs in hand-written gfx950 kernels in csrc/libkeds_hip.so.
"""
from ._lib import LIB_PATH, build as build_library, load as load_library   # noqa: F401
from .index import (FlatIndex, IndexFlatIP, IndexFlatL2, PackedExchange, ShardedFlatIndex, exchange_merge_gather,   # noqa: F401
                    index_cpu_to_all_gpus, merge_partials, shard_bounds)
from .model import (CLIP, CrossAttention, CrossFormer, IM2TEXT, KnowledgeStream, LayerNorm, QuickGELU,   # noqa: F401
                    ResidualAttentionBlock, Transformer, VisualTransformer, build_model,
                    clip_config_from_state_dict, convert_models_to_fp32, convert_weights)
from .retrieval import (all_gather_features, build_database, compose_query_features, extract_feature_database, extract_feature_database_sharded, load_database_shard, get_cirr_testoutput, get_metrics_cirr, get_metrics_cirr_topk,   # noqa: F401
                        get_metrics_coco, get_metrics_fashion, get_metrics_imgnet, get_retrieved_features,
                        load_checkpoint, make_stream_modules)

from . import clip   # noqa: F401  (load / _transform / tokenize of the reference's model/clip.py)
from .clip import load, tokenize   # noqa: F401

__version__ = "0.1.0"
