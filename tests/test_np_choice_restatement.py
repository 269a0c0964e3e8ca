"""The arithmetic the device-side anti-noise sampling implements (cim_amd/csrc/mining.hip, launch 3), restated
in Python scalar by scalar and checked against NumPy itself: np.random.choice(a, n, True, p) == searchsorted of
the normalised f64 cumsum at the next n doubles of the legacy MT19937 stream, with p = prob / prob.sum() in f32
where .sum() is NumPy's pairwise summation (128-element blocks, 8 accumulators, recursion on halves above that) -
reference lib/modeling/heads.py:457-461.  The GPU tests check the kernel against the reference goldens; this
test pins the restatement (including the > 128-element branch the goldens do not reach) without a GPU."""
import numpy as np

f32 = np.float32


def pairwise_sum(a):
    n = len(a)
    if n < 8:
        r = f32(0)
        for x in a:
            r = f32(r + x)
        return r
    if n <= 128:
        r = [f32(a[i]) for i in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = f32(r[j] + a[i + j])
            i += 8
        res = f32(f32(f32(r[0] + r[1]) + f32(r[2] + r[3])) + f32(f32(r[4] + r[5]) + f32(r[6] + r[7])))
        while i < n:
            res = f32(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f32(pairwise_sum(a[:n2]) + pairwise_sum(a[n2:]))


def test_choice_restatement_matches_numpy():
    rng = np.random.RandomState(0)
    sizes = list(range(1, 40)) + [127, 128, 129, 135, 200, 255, 256, 257, 263, 519, 1000, 1023, 1024]
    for trial, n in enumerate(sizes + [int(rng.randint(1, 1025)) for _ in range(150)]):
        prob = (rng.rand(n).astype(f32) ** 3 * f32(10 ** rng.uniform(-6, 0))).astype(f32) + f32(1e-12)
        total = pairwise_sum(prob)
        assert total == prob.sum(), n
        p32 = np.array([f32(np.float64(x) / np.float64(total)) for x in prob], dtype=f32)   # f64 divide + one rounding
        assert np.array_equal(p32, prob / prob.sum())
        cdf = np.zeros(n)
        acc = 0.0
        for j in range(n):
            acc = float(p32[j]) if j == 0 else acc + float(p32[j])
            cdf[j] = acc
        cdf = cdf / cdf[-1]
        np.random.seed(trial)
        state = np.random.get_state()
        ref = np.random.choice(np.arange(n), size=n, replace=True, p=prob / prob.sum())
        probe = np.random.random_sample()
        np.random.set_state(state)                      # what heads._RngLedger does: snapshot, draw, rewind, re-draw
        u = np.random.random_sample(n + 17)[:n]
        np.random.set_state(state)
        np.random.random_sample(n)
        assert np.array_equal(ref, np.searchsorted(cdf, u, side="right"))
        assert np.random.random_sample() == probe
