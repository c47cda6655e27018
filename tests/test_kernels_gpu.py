"""-m gpu: every HIP kernel family, called through the C ABI, against its torch restatement."""
import pytest

import kernel_checks as kc

pytestmark = pytest.mark.gpu

FAMILIES = ["probe", "gemm", "gemm_races", "conv", "attention", "norms", "streaming", "wo", "image_prep"]


@pytest.mark.parametrize("family", FAMILIES)
def test_kernel_family(hip_env, family):
    hip, emu, dev, ops = hip_env
    fn = dict(kc.all_checks(hip, emu, dev, ops))[family]
    results = fn()
    bad = [(n, e, t) for n, e, t in results if not (e <= t)]
    assert not bad, "parity failures:\n" + "\n".join(f"  {n}: rel_l2={e:.3e} > tol={t:.1e}" for n, e, t in bad)
