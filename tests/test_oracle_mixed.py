"""Oracle: the general mixed-model path -- random slopes, crossed and nested extra groupings.

Data generation follows data_gen.rs:146-284 (draw order: primary intercepts, primary slopes, all extra
intercepts, all extra slopes) and :812-842 (latent-scale contribution); level ids follow
engine-contract/src/generation.rs:297-427.  Pinned bit-exact to the 11 multi-grouping / slope datasets the
reference captured at sim 0 (validation/data/lmm_*.rds, formulas.R:382-539)."""
import numpy as np
import pytest

from helpers import load_case, mixed_case_names


@pytest.mark.parametrize("name", mixed_case_names())
def test_mixed_model_datasets_are_reproduced_bit_exact(name, oracle):
    spec, base_seed, a, low = load_case(name)
    x, y, ids, extra = oracle.generate_sim_data_ids(spec, base_seed, 0, low["n"])
    assert np.array_equal(ids, a["cluster_ids"])
    assert len(extra) == a["extra_ids"].shape[0]
    for g, e in enumerate(extra):
        assert np.array_equal(e, a["extra_ids"][g]), g
    assert np.array_equal(x, a["design"])
    assert np.array_equal(y, a["outcome"])
