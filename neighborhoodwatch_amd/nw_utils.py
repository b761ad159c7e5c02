"""On-disk naming contract of the hot path + the model-name -> dimension table the CLI needs.

Mirrors the parts of the reference's ``neighborhoodwatch/nw_utils.py`` and ``model_generator.py`` that the kNN
path and its callers touch; the file names ARE the API between the kNN stage and the merge stage
(nw_utils.py:94-107, merge.py:15-27).  Everything that needs the network (``check_dataset_exists_remote``,
nw_utils.py:18-23; the embedding vendors of model_generator.py) is out of scope.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np

# model name -> default embedding width (model_generator.py:26-96).  Only the table survives: `nw -m NAME` derives
# d from it (neighborhoodwatch.py:98); the generators behind the names need remote services.
MODEL_DIMENSIONS = {
    "text-embedding-ada-002": 1536,
    "text-embedding-3-small": 1536,
    "text-embedding-3-large": 3072,
    "textembedding-gecko@003": 768,
    "text-embedding-004": 768,
    "text-embedding-005": 768,
    "intfloat/e5-large-v2": 1024,
    "intfloat/e5-base-v2": 768,
    "intfloat/e5-small-v2": 384,
    "nvidia-nemo": 1024,
    "cohere/embed-english-v3.0": 1024,
    "cohere/embed-english-light-3.0": 384,
    "voyage-3-large": 1024,
    "voyage-3-lite": 512,
    "colbertv2.0": 128,
}
_RESIZABLE_UP_TO_DEFAULT = ("text-embedding-3-small", "text-embedding-3-large")   # model_generator.py:103-105
_VOYAGE_LARGE = "voyage-3-large"
_VOYAGE_LARGE_SIZES = (256, 512, 1024, 2048)                                       # model_generator.py:106-108


def get_valid_model_name_list():
    return list(MODEL_DIMENSIONS)


def get_valid_model_names_string() -> str:
    return ", ".join(MODEL_DIMENSIONS)


def is_valid_model_name(model_name) -> bool:
    return model_name in MODEL_DIMENSIONS


def get_default_model_dimension_size(model_name: str) -> int:
    if not is_valid_model_name(model_name):
        raise AssertionError(f"invalid model name {model_name!r}; must be one of: {get_valid_model_names_string()}")
    return MODEL_DIMENSIONS[model_name]


def get_effective_embedding_size(model_name: str, output_dimension_size: int = None) -> int:
    """model_generator.py:99-113: an explicit size is honoured only for the resizable models."""
    default = get_default_model_dimension_size(model_name)
    if output_dimension_size is None:
        return default
    if model_name in _RESIZABLE_UP_TO_DEFAULT:
        assert output_dimension_size <= default
        return output_dimension_size
    if model_name == _VOYAGE_LARGE:
        assert output_dimension_size in _VOYAGE_LARGE_SIZES
        return output_dimension_size
    return default


def output_dimension_validity_check(model_name: str, given_dimension: int, actual_dimension: int) -> bool:
    """nw_utils.py:157-164: column count must equal d (voyage-3-large binary packs 8 dims per column)."""
    if actual_dimension == given_dimension:
        return True
    return model_name == _VOYAGE_LARGE and given_dimension == 8 * actual_dimension


# ---- paths ----------------------------------------------------------------------------------------------
def get_full_filename(data_dir, filename) -> str:
    """nw_utils.py:26-30: prefix with data_dir unless already prefixed."""
    data_dir, filename = str(data_dir), str(filename)
    return filename if filename.startswith(data_dir) else f"{data_dir}/{filename}"


def get_model_prefix(model_name) -> str:
    return model_name.replace("/", "_") if model_name else "text-embedding-ada-002"      # nw_utils.py:33-38


def get_model_data_homedir(output_homedir, model_name, query_count, base_count, k) -> str:
    return f"{output_homedir}/{get_model_prefix(model_name)}/q{query_count}_b{base_count}_k{k}"    # nw_utils.py:62-64


def setup_model_output_folder(output_homedir, model_name, query_count, base_count, k) -> str:
    data_dir = get_model_data_homedir(output_homedir, model_name, query_count, base_count, k)
    os.makedirs(f"{data_dir}/partial", exist_ok=True)                                      # nw_utils.py:67-73
    return data_dir


def _source_name(kind, model_name, row_count, output_dimension, output_dtype) -> str:
    parts = [model_name.replace("/", "_"), str(output_dimension)]
    if output_dtype is not None:
        parts.append(str(output_dtype))
    parts += [f"{kind}_vector_data", str(row_count)]
    return "_".join(parts) + ".parquet"


def get_source_query_dataset_filename(knn_model_data_homedir, model_name, row_count, output_dimension=None, output_dtype=None) -> str:
    return f"{knn_model_data_homedir}/{_source_name('query', model_name, row_count, output_dimension, output_dtype)}"   # nw_utils.py:76-82


def get_source_base_dataset_filename(knn_model_data_homedir, model_name, row_count, output_dimension=None, output_dtype=None) -> str:
    return f"{knn_model_data_homedir}/{_source_name('base', model_name, row_count, output_dimension, output_dtype)}"    # nw_utils.py:85-91


def _partial_name(stem: str, knn_model_data_homedir, partial_set_cnt: int) -> str:
    tail = f"final_{stem}" if partial_set_cnt == -1 else f"{stem}{partial_set_cnt}"        # -1 selects the final file
    return f"{knn_model_data_homedir}/partial/{tail}.parquet"


def get_partial_indices_filename(knn_model_data_homedir, partial_set_cnt: int) -> str:
    return _partial_name("indices", knn_model_data_homedir, partial_set_cnt)               # nw_utils.py:94-99


def get_partial_distances_filename(knn_model_data_homedir, partial_set_cnt: int) -> str:
    return _partial_name("distances", knn_model_data_homedir, partial_set_cnt)             # nw_utils.py:102-107


# ---- small vector helpers used by the dataset writers / tests ---------------------------------------------
def is_zero_embedding(embedding) -> bool:
    return not np.any(embedding)


def normalize_vector(vector):
    """nw_utils.py:56-59."""
    assert not is_zero_embedding(vector), "Zero vector found!"
    return (vector / np.linalg.norm(vector)).astype(np.float32)
