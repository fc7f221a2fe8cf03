"""Pin the CPU oracle's RNG / transform plane to the reference's own vectors
(crates/engine-core/tests/rng_byte_identity.rs:18-50, golden_rng.rs:37-235),
restated as data in tests/golden/rng_kat.json."""
import numpy as np
import pytest

from helpers import load_kat, splitmix64

KAT = load_kat()


def test_philox_known_answers(oracle):
    for v in KAT["philox"]:
        assert oracle.philox4x32_10(v["ctr"], v["key"]) == v["out"]


def test_splitmix64_pinned(oracle):
    for z, want in KAT["splitmix64"]:
        assert oracle.splitmix64(z) == want
        assert splitmix64(z) == want  # the helper the tests derive base_seed with


def test_u32_to_unit_f32_bits(oracle):
    for w, bits in KAT["u32_to_unit_f32_bits"]:
        assert oracle.u32_to_unit_f32_bits(w) == bits


@pytest.mark.parametrize("kind,sim,key", [("normal", 0, "normals_seed42_sim0_bits"),
                                          ("uniform", 1, "uniforms_seed42_sim1_bits"),
                                          ("scenario_uniform", 0, "scenario_uniforms_seed42_sim0_bits")])
def test_stream_goldens_bitwise(oracle, kind, sim, key):
    got = oracle.stream_f32(42, sim, kind, 64).view(np.uint32)
    assert got.tolist() == KAT[key]


def test_categorical_golden(oracle):
    got = oracle.stream_categorical(42, 2, KAT["categorical_probs"], 64)
    assert got.tolist() == KAT["categoricals_seed42_sim2"]


def test_phi_and_marginal_uniform_goldens(oracle):
    # The reference composes glmm's owned exp (<= 1 ULP of libm, distributions.rs:21-23); the oracle uses
    # libm, so the pin is absolute 2^-52 on phi, not bitwise.
    z = np.array(KAT["z_grid"])
    want_phi = np.array(KAT["phi_bits"], dtype=np.uint64).view(np.float64)
    want_mu = np.array(KAT["marginal_uniform_bits"], dtype=np.uint64).view(np.float64)
    s3 = 1.7320508075688772
    for zi, wp, wm in zip(z, want_phi, want_mu):
        assert abs(oracle.phi(float(zi)) - wp) <= 2.0 ** -52
        assert abs(oracle.lib().mcpo_marginal_uniform(float(zi), -s3, s3) - wm) <= 2.0 ** -50


def test_icdf_matches_scipy_within_reference_gate(oracle):
    # rng.rs:272-273: validated max |dz| = 3.0e-4 over the reachable range (gate 2e-3)
    from scipy.stats import norm
    u = (2 * np.arange(0, 2 ** 23, 4099, dtype=np.int64) + 1).astype(np.float64) * 2.0 ** -24
    z = oracle.norm_inv_cdf_f32(u.astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(z - norm.ppf(u))) < 2e-3


def test_planar_fill_matches_scalar_contract(oracle):
    # rng.rs:230-231: fill_normal_column == norm_inv_cdf_f32(u32_to_unit_f32(word(class,col,row)))
    key = [0x12345678, 0x9ABCDEF0]
    col = oracle.fill_normal_column(key, 1, 3, 37)
    for i in (0, 1, 5, 36):
        w = oracle.philox4x32_10([i >> 2, 3, 1, 0], key)[i & 3]
        u = np.uint32(oracle.u32_to_unit_f32_bits(w)).view(np.float32)
        assert col[i].view(np.uint32) == oracle.norm_inv_cdf_f32(np.array([u]))[0].view(np.uint32)
    # prefix stability in max_n (rng.rs:604)
    assert np.array_equal(oracle.fill_normal_column(key, 1, 3, 200)[:37].view(np.uint32), col.view(np.uint32))
