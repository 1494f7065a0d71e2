"""GPU tests of the in-kernel noise (``rng="philox"``: the default and the benchmarked mode of the walks).

The reference draws ``torch.randn_like(y)`` per integrator step (``/root/reference/src/jamun/sampling/mcmc/functional/_splitting.py:161``,
``:93`` for ABOBA): iid N(0,1) per atom and component, a different stream per rank (``cmdline/sample.py:86-88``: seed + rank).  The
HIP path replaces it by Philox4x32-10 keyed by (seed, iteration, atom) + Box-Muller, exported as ``jamun_philox_normal``.  Checked
here: (a) the counter-based generator against the published known-answer vectors and the draws against a float64 restatement,
(b) moments and correlations over 4e7 draws, (c) stream separation by seed, (d) that the fused walks consume exactly these draws.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def philox4x32_10(ctr, key):
    """Philox4x32 with 10 rounds (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) on numpy
    arrays: ctr = 4 uint32 words, key = 2 uint32 words -> 4 uint32 words."""
    c = [np.asarray(x, dtype=np.uint64) for x in ctr]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    m0, m1, mask, s32 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF), np.uint64(32)
    for _ in range(10):
        p0, p1 = m0 * c[0], m1 * c[2]
        c = [(p1 >> s32) ^ c[1] ^ k0, p1 & mask, (p0 >> s32) ^ c[3] ^ k1, p0 & mask]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return [x.astype(np.uint32) for x in c]


def reference_normals(seed, iteration, atoms):
    """[len(atoms), 3] draws as include/jamun_hip.h documents them: counter (atom, iteration, 0x4a414d55, 0), key = seed words,
    u = (float32(word) + 0.5) * 2^-32, Box-Muller.  float32 where the contract says float32 (the uniforms and the angle), float64
    for the transcendental functions."""
    atoms = np.asarray(atoms, dtype=np.uint64)
    w = philox4x32_10([atoms, np.full_like(atoms, iteration), np.full_like(atoms, 0x4A414D55), np.zeros_like(atoms)],
                      (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u = [np.minimum((x.astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -32), np.float32(1.0)) for x in w]
    two_pi = np.float32(6.283185307179586)
    r0 = np.sqrt(-2.0 * np.log(np.maximum(u[0], np.float32(1e-10)).astype(np.float64)))
    r1 = np.sqrt(-2.0 * np.log(np.maximum(u[2], np.float32(1e-10)).astype(np.float64)))
    a0, a1 = (two_pi * u[1]).astype(np.float64), (two_pi * u[3]).astype(np.float64)
    return np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1)], axis=1)


def test_philox_reference_implementation_known_answers():
    """Random123's kat_vectors for philox4x32 with 10 rounds pin the restatement the kernel is compared with."""
    kat = [((0, 0, 0, 0), (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), "d16cfe09 94fdcceb 5001e420 24126ea1")]
    for ctr, key, want in kat:
        assert " ".join(f"{int(x):08x}" for x in philox4x32_10(ctr, key)) == want


def test_kernel_draws_match_the_documented_generator():
    from jamun_amd import native

    dev = torch.device("cuda", 0)
    worst = 0.0
    for seed, it, first, n in [(0, 1, 0, 4096), (1234, 7, 0, 5000), (42 + (1 << 40), 19999, 1_000_000, 3000), (2**64 - 1, 2**32 - 1, 2**32 - 100, 100),
                               (7, 0, 17, 1)]:
        got = native.philox_normal(n, seed, it, dev, first_atom=first).cpu().double().numpy()
        atoms = (np.arange(n, dtype=np.uint64) + np.uint64(first)) & np.uint64(0xFFFFFFFF)
        ref = reference_normals(seed, it, atoms)
        worst = max(worst, float(np.abs(got - ref).max()))
        assert np.isfinite(got).all()
    assert worst <= 1e-6, worst  # logf / sqrtf / cosf / sinf of the device against float64


def test_noise_moments_and_correlations():
    """iid N(0,1) per atom, component and iteration (the contract of torch.randn_like): 1.33e7 atoms x 3 components x 2
    consecutive iterations; standard errors are 1.6e-4 (mean), 2.2e-4 (variance), 2.7e-4 (correlations)."""
    from jamun_amd import native

    dev = torch.device("cuda", 0)
    n, seed = 13_333_334, 1234
    a = native.philox_normal(n, seed, 5, dev).double()
    b = native.philox_normal(n, seed, 6, dev).double()
    assert torch.isfinite(a).all() and a.abs().max() < 7.0  # u >= 2^-33: |R| <= sqrt(-2 ln 2^-33) = 6.76
    flat = a.reshape(-1)
    assert flat.numel() >= 4e7
    mean, var = flat.mean().item(), flat.var().item()
    assert abs(mean) < 1e-3 and abs(var - 1.0) < 1e-3, (mean, var)
    assert abs((flat ** 3).mean().item()) < 5e-3 and abs((flat ** 4).mean().item() - 3.0) < 1e-2  # skewness, kurtosis
    tail = (flat.abs() > 3.0).double().mean().item()
    assert abs(tail - 0.0026998) < 1e-4, tail
    for c in range(3):
        m, v = a[:, c].mean().item(), a[:, c].var().item()
        assert abs(m) < 1e-3 and abs(v - 1.0) < 1.5e-3, (c, m, v)

    def corr(x, y):
        x, y = x - x.mean(), y - y.mean()
        return ((x * y).mean() / (x.std() * y.std())).item()

    pairs = {"components 0,1": (a[:, 0], a[:, 1]), "components 0,2": (a[:, 0], a[:, 2]), "components 1,2": (a[:, 1], a[:, 2]),
             "neighbouring atoms": (a[:-1, 0], a[1:, 0]), "neighbouring atoms, z": (a[:-1, 2], a[1:, 2]),
             "atoms 17 apart (next walker of the bench batch)": (a[:-17, 1], a[17:, 1]),
             "consecutive iterations": (a[:, 0], b[:, 0]), "consecutive iterations, z": (a[:, 2], b[:, 2]),
             "squares of the Box-Muller pair": (a[:, 0] ** 2, a[:, 1] ** 2)}
    for name, (x, y) in pairs.items():
        r = corr(x, y)
        assert abs(r) < 1.2e-3, (name, r)


def test_streams_of_different_seeds_and_ranks_differ():
    """cmdline/sample.py:86-88 seeds every rank with seed + rank: the streams must be unrelated, not shifted copies."""
    from jamun_amd import native

    dev = torch.device("cuda", 0)
    n = 2_000_000
    s0 = native.philox_normal(n, 42, 3, dev).double()
    s1 = native.philox_normal(n, 43, 3, dev).double()
    hi = native.philox_normal(n, 42 + (1 << 32), 3, dev).double()  # differs in the high key word only
    again = native.philox_normal(n, 42, 3, dev).double()
    assert torch.equal(s0, again)
    for other in (s1, hi):
        assert (s0 == other).double().mean().item() < 1e-6
        r = ((s0 - s0.mean()) * (other - other.mean())).mean().item() / (s0.std() * other.std()).item()
        assert abs(r) < 2e-3, r
        # not a shifted copy either (atom i of one stream vs atom i + 1 of the other)
        r = (s0[:-1] * other[1:]).mean().item()
        assert abs(r) < 2e-3, r
    # first_atom continues the same stream: a rank-local batch sees atoms first .. first + n - 1
    part = native.philox_normal(1000, 42, 3, dev, first_atom=5000).double()
    assert torch.equal(part, s0[5000:6000])


@pytest.mark.parametrize("integrator", ["baoab", "aboba"])
def test_walks_without_a_noise_tensor_consume_exactly_these_draws(integrator):
    """jamun_walk_* with noise_dev == NULL is bit-identical to the same walk fed jamun_philox_normal(seed, iteration) as its noise
    tensor — so the distribution tests above are tests of the benchmarked mode."""
    from jamun_amd import native, synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    dev = torch.device("cuda", 0)
    mols = [synth.random_chain(17, seed=0)] * 3 + [synth.random_chain(9, seed=1)] * 2
    batch = WalkerBatch.from_molecules(mols).to(dev)
    model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint(output_gain=0.1)).to(dev)
    smp = model.sampler_for(batch, 0.04)
    steps, seed, n = 7, 977 + (3 << 33), batch.num_nodes
    params = native.make_mcmc_params(steps, 0.04, 1.0, 1.0, 1.0, 100.0)
    torch.manual_seed(0)
    y0 = batch.pos + 0.04 * torch.randn_like(batch.pos)
    v0 = torch.randn_like(y0)
    ya, va = y0.clone(), v0.clone()
    out_a = smp.walk(integrator, ya, va, params, None, seed, True)
    noise = torch.stack([native.philox_normal(n, seed, i, dev) for i in range(1, steps)])
    yb, vb = y0.clone(), v0.clone()
    out_b = smp.walk(integrator, yb, vb, params, noise, 0, True)
    assert torch.equal(ya, yb) and torch.equal(va, vb)
    for a, b in zip(out_a, out_b):
        assert (a is None and b is None) or torch.equal(a, b)
    # and the noise matters: another seed gives another trajectory
    yc, vc = y0.clone(), v0.clone()
    smp.walk(integrator, yc, vc, params, None, seed + 1, True)
    assert (yc - ya).abs().max().item() > 1e-4
