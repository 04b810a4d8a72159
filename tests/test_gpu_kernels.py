"""GPU parity tests, kernel level: every C-ABI block of include/mlsp_hip.h against the oracle / a plain
torch fp32 restatement of the same op on identical seeded inputs.  Run with `-m gpu` on an MI355X."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_common as gc
from oracle import knn_canon, ref_cpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mlsp_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _fh():
    from mlsp_amd import functional as Fh
    return Fh


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


# ----------------------------------------------------------------------------- kNN: bit-exact
@pytest.mark.parametrize("B,N,C,k", [(2, 256, 3, 20), (2, 1024, 3, 20), (2, 256, 64, 20), (1, 1024, 128, 20),
                                     (3, 20, 3, 20), (2, 33, 64, 20), (2, 100, 5, 7), (1, 300, 3, 40), (2, 70, 16, 1),
                                     # v5 (two waves per query group) paths: LDS-resident / streamed, vector / scalar loads
                                     (2, 384, 16, 20), (2, 256, 5, 32), (1, 512, 100, 20), (1, 2048, 3, 20),
                                     (1, 4096, 16, 8), (2, 640, 64, 32), (8, 128, 64, 20), (9, 256, 128, 20),
                                     # the benchmarked shapes themselves (BASELINE.json configs[1]: B = 32, N = 1024; XCD-mapped 8-wave path)
                                     (32, 1024, 3, 20), (32, 1024, 64, 20), (32, 1024, 128, 20), (32, 2048, 64, 20),
                                     # k > 32 (configs[4]: k = 40): the two-pass select with 128 chunk maxima per query
                                     (2, 2048, 3, 40), (8, 2048, 64, 40), (3, 1024, 128, 40), (2, 512, 64, 33), (1, 256, 16, 48),
                                     (1, 128, 64, 64), (2, 2048, 64, 64), (2, 1024, 64, 25), (1, 384, 100, 40),
                                     # v6 (knn6.hip: k <= 24, N % 128 == 0): fewest tiles, ragged tile-ring tails, k = 1 / 24, unaligned rows
                                     (1, 128, 3, 1), (2, 128, 64, 24), (2, 256, 3, 24), (3, 384, 5, 20), (2, 1152, 64, 20), (1, 1152, 3, 20),
                                     (2, 640, 33, 7), (1, 2048, 128, 24), (16, 128, 16, 20),
                                     # v6 wide (knn6w_kernel: 24 < k <= 40, N % 128 == 0, C > 16): one tile per quarter, ragged rings, the configs[4] shape
                                     (2, 128, 64, 40), (3, 256, 64, 25), (2, 640, 128, 40), (1, 1152, 33, 37), (16, 2048, 64, 40), (1, 4096, 64, 40),
                                     # the VALU kernel: 32 < k <= 40 on shapes outside the two-pass kernel (ragged N, N < 128, 64 < C < 128, C > 128)
                                     (2, 150, 200, 40), (1, 96, 200, 36), (2, 100, 24, 33)])
def test_knn_bit_exact_vs_oracle(dev, B, N, C, k):
    Fh = _fh()
    xp = _rand((B * N, C), 100 + N + C)
    want = knn_canon.knn_point_major(xp.view(B, N, C), k)
    g = Fh.knn_graph(xp.to(dev), B, N, k)
    got = g.idx.view(B, N, k).cpu().numpy()
    assert np.array_equal(got, want), "mismatching rows: %d" % int((got != want).any(-1).sum())
    # reverse index: every (i, slot) appears exactly once under its destination, sorted
    off = g.rev_off.cpu().numpy()
    ent = g.rev_ent.cpu().numpy()
    assert off[0] == 0 and off[-1] == B * N * k and np.all(np.diff(off) >= 0)
    dest = np.repeat(np.arange(B * N), np.diff(off))
    src = (dest // N) * N + (ent >> 8)
    slot = ent & 255
    assert np.array_equal(want.reshape(B * N, k)[src, slot] + (dest // N) * N, dest)
    key = dest.astype(np.int64) * (1 << 40) + ent
    assert np.all(np.diff(key) > 0)


def test_knn_duplicates_and_strided_input(dev):
    Fh = _fh()
    x = torch.zeros(1, 64, 3)
    x[0, :, 0] = torch.arange(64) // 2            # pairs of identical points -> ties resolved by index
    want = knn_canon.knn_point_major(x, 20)
    got = Fh.knn_graph(x.view(64, 3).to(dev), 1, 64, 20).idx.view(1, 64, 20).cpu().numpy()
    assert np.array_equal(got, want)
    # row-strided view (a column slice of a wider matrix) is read in place
    wide = _rand((2 * 128, 96), 5).to(dev)
    sl = wide[:, 32:96]
    got = Fh.knn_graph(sl, 2, 128, 20).idx.view(2, 128, 20).cpu().numpy()
    want = knn_canon.knn_point_major(sl.cpu().contiguous().view(2, 128, 64), 20)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("k", [20, 40])
@pytest.mark.parametrize("C,scale", [(3, 1e-6), (64, 1e-6), (3, 3e-4), (128, 1e-5), (64, 0.0)])
def test_knn_near_duplicates_need_exact_distances(dev, C, scale, k):
    """v6 ranks survivors by split-bf16 distances and recomputes the canonical fp32 distance only where two survivors are closer than
    the error bound.  Clouds made of 32 tight clusters (copies of a point + noise of 1e-6 .. 3e-4, or exact copies) put MANY candidates
    inside that bound -- some queries overflow their survivor lists (exact path), the rest resolve dozens of ambiguous pairs per query:
    indices must still be bit-exact.  k = 40: the wide kernel (knn6w_kernel) with the v5 kernel behind it on the clouds whose lists
    overflowed."""
    Fh = _fh()
    B, N = 3, 512
    g = torch.Generator().manual_seed(77 + C)
    centres = torch.rand(B, 32, C, generator=g) * 2 - 1
    x = centres[:, torch.arange(N) % 32, :] + scale * torch.randn(B, N, C, generator=g)
    want = knn_canon.knn_point_major(x, k)
    got = Fh.knn_graph(x.view(B * N, C).to(dev), B, N, k).idx.view(B, N, k).cpu().numpy()
    assert np.array_equal(got, want), "mismatching rows: %d" % int((got != want).any(-1).sum())


@pytest.mark.parametrize("k", [20, 40])
@pytest.mark.parametrize("C,offset,B,N", [(64, 3.0, 4, 1024), (128, 1.5, 2, 1024), (64, 50.0, 2, 512), (3, 20.0, 2, 1024)])
def test_knn_clouds_far_from_the_origin(dev, C, offset, B, N, k):
    """Graph-stage features sit far from the origin compared with their spread (BatchNorm + LeakyReLU + max over k).  v6 sweeps the
    cloud in coordinates relative to its first point (error of the split products ~ the centred norms) and budgets the canonical
    arithmetic's own rounding on the raw coordinates separately; with a large offset the canonical fp32 distances are coarse (many exact
    ties): indices must still be bit-exact."""
    Fh = _fh()
    x = _rand((B, N, C), 300 + C) * 0.5 + offset
    want = knn_canon.knn_point_major(x, k)
    got = Fh.knn_graph(x.view(B * N, C).to(dev), B, N, k).idx.view(B, N, k).cpu().numpy()
    assert np.array_equal(got, want), "mismatching rows: %d" % int((got != want).any(-1).sum())


def test_knn_nonfinite_rows_do_not_poison_the_others(dev):
    """A point with an infinite coordinate makes every bound of its cloud infinite: the exact path decides, and the finite points still
    get their canonical neighbours among the finite candidates (a NaN / inf distance is never selected, oracle/knn_canon.c)."""
    Fh = _fh()
    B, N, C, k = 2, 256, 3, 20
    x = _rand((B, N, C), 91)
    x[1, 7, 0] = float("inf")
    want = knn_canon.knn_point_major(x, k)
    got = Fh.knn_graph(x.view(B * N, C).to(dev), B, N, k).idx.view(B, N, k).cpu().numpy()
    assert np.array_equal(got[0], want[0])
    finite = np.ones(N, bool)
    finite[7] = False
    assert np.array_equal(got[1][finite], want[1][finite])


@pytest.mark.parametrize("C,k", [(3, 20), (64, 20), (128, 20), (64, 40), (3, 40)])
@pytest.mark.parametrize("where", ["query", "first_point", "whole_cloud", "most_of_cloud"])
def test_knn_nan_rows_are_defined_and_in_range(dev, C, k, where):
    """Features go NaN when a training run diverges; the reference then propagates NaN, it does not read out of bounds.  Every index the
    kernels return must lie in [0, N) -- the gathers downstream do not range-check -- and must equal the oracle's: a NaN distance is
    never selected, a row with fewer than k comparable candidates ends in zeros (oracle/knn_canon.c), and the finite rows of a cloud
    that holds a NaN point keep their canonical neighbours.  `first_point`: the origin of v6's centred image is NaN (every approximate
    distance of the cloud is NaN: the exact path has to decide)."""
    Fh = _fh()
    B, N = 3, 256
    x = _rand((B, N, C), 400 + C)
    nan = float("nan")
    if where == "query":
        x[1, 37, C // 2] = nan
    elif where == "first_point":
        x[1, 0, 0] = nan
    elif where == "whole_cloud":
        x[1] = nan
    else:                                                     # only 11 comparable candidates left (< k): rows end in zeros
        x[1, 11:, 0] = nan
    want = knn_canon.knn_point_major(x, k)
    idx = Fh.knn_graph(x.view(B * N, C).to(dev), B, N, k).idx
    got = idx.view(B, N, k).cpu().numpy()
    assert got.min() >= 0 and got.max() < N
    assert np.array_equal(got, want), "mismatching rows: %d" % int((got != want).any(-1).sum())


@pytest.mark.parametrize("k", [20, 28, 40, 64])
def test_knn_massive_ties_overflow_path(dev, k):
    """Every survivor buffer overflows (all / many candidates tie): the exact sequential-insertion pass decides, ties -> lower index.
    k = 28..64 run the two-entries-per-lane lists of the k > 24 kernel."""
    Fh = _fh()
    N = 256
    same = torch.zeros(2, N, 16)
    same[1] = 0.5                                             # two clouds of N identical points
    got = Fh.knn_graph(same.view(2 * N, 16).to(dev), 2, N, k).idx.view(2, N, k).cpu().numpy()
    assert np.array_equal(got, np.broadcast_to(np.arange(k), (2, N, k)))
    x = torch.zeros(1, N, 3)
    x[0, :, 0] = (torch.arange(N) // 64).float()              # four clusters of 64 duplicates
    x[0, :, 1] = (torch.arange(N) % 2).float() * 1e-3         # ... split in two interleaved sub-clusters
    want = knn_canon.knn_point_major(x, k)
    got = Fh.knn_graph(x.view(N, 3).to(dev), 1, N, k).idx.view(1, N, k).cpu().numpy()
    assert np.array_equal(got, want)


def test_knn_golden_reference_rows(dev, golden_dir):
    """Against the reference's own indices (fixture): identical on every unambiguous row."""
    Fh = _fh()
    for f in sorted(os.listdir(golden_dir)):
        if not f.startswith("knn_"):
            continue
        g = dict(np.load(os.path.join(golden_dir, f)))
        x = torch.from_numpy(g["x"])
        B, C, N = x.shape
        k = g["idx"].shape[-1]
        got = Fh.knn_graph(x.transpose(2, 1).contiguous().view(B * N, C).to(dev), B, N, k).idx.view(B, N, k).cpu().numpy()
        eps = 8e-6 * C
        safe = (g["gap_k"] > eps) & (g["gap_in"] > eps)
        # the comparison has to mean something: the rows whose rank gaps exceed the reference's own fp32 noise are (nearly) all rows
        # (the six fixtures: 92.6 % .. 100 % of their rows at this conservative eps)
        assert safe.mean() >= 0.92, (f, float(safe.mean()))
        assert np.array_equal(got[safe], g["idx"][safe]), f
        # over ALL rows, the ambiguous ones included, the canonical arithmetic reproduces the reference's ordered rows almost everywhere
        # (measured: every row of every fixture)
        same = (got == g["idx"]).all(-1).mean()
        assert same >= 0.99, (f, float(same))
        # ... and on the others the neighbour SET is still the reference's unless the k / k+1 gap itself is inside the noise
        amb = ~safe & (g["gap_k"] > eps)
        assert all(set(a) == set(b) for a, b in zip(got[amb].tolist(), g["idx"][amb].tolist())), f


def test_knn_public_api_and_errors(dev):
    from mlsp_amd import model_utils as mu
    x = _rand((2, 3, 128), 1).to(dev)
    idx = mu.knn(x, 20)
    assert idx.dtype == torch.int64 and idx.shape == (2, 128, 20)
    assert torch.equal(idx[:, :, 0].cpu(), torch.arange(128).expand(2, 128))       # self is the nearest
    with pytest.raises(RuntimeError):
        mu.knn(x[:, :, :10], 20)                                                    # k > N
    with pytest.raises(RuntimeError):
        mu.knn(x.cpu(), 20)                                                         # no CPU fallback


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 1, 300, 200, 64), (0, 1, 1024, 128, 3), (0, 0, 257, 130, 100),
                                         (1, 0, 64, 6, 5000), (1, 0, 256, 512, 4096), (0, 1, 32, 512, 1024),
                                         (0, 1, 4096, 2048, 512), (1, 1, 100, 70, 50), (0, 1, 5, 9, 256),
                                         # skinny.hip: <= 32 rows (fwd / dgrad) and 32-deep wgrads, ragged edges
                                         (0, 1, 32, 256, 1024), (0, 1, 17, 33, 100), (0, 0, 32, 1024, 256), (0, 0, 7, 50, 9),
                                         (1, 0, 256, 1024, 32), (1, 0, 9, 256, 20), (1, 0, 70, 33, 3)])
def test_gemm(dev, ta, tb, M, N, K):
    Fh = _fh()
    A = _rand((K, M) if ta else (M, K), 1)
    B = _rand((N, K) if tb else (K, N), 2)
    want = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
    got = Fh.gemm(A.to(dev), B.to(dev), ta=bool(ta), tb=bool(tb)).cpu().double()
    err = (got - want).abs().max().item()
    assert err < 2e-6 * K ** 0.5 * 4 + 1e-6, err


@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 1, 1024, 256, 512), (0, 0, 512, 384, 256), (1, 0, 256, 128, 8192), (0, 1, 4096, 128, 64),
                                         (1, 1, 256, 256, 128), (0, 1, 300, 200, 64)])
def test_gemm_bf16_operand_mode(dev, ta, tb, M, N, K):
    """gemm_precision("bf16") (MLSP_PREC_BF16 per call): operands rounded to bf16 (RNE), fp32 accumulation -- compared with that exact model in
    float64; shapes off the fast path (last case) keep using the fp32 kernel.  The switch is restored afterwards."""
    Fh = _fh()
    A = _rand((K, M) if ta else (M, K), 11)
    B = _rand((N, K) if tb else (K, N), 12)
    opA, opB = (A.t() if ta else A), (B.t() if tb else B)
    with Fh.gemm_precision("bf16"):
        got = Fh.gemm(A.to(dev), B.to(dev), ta=bool(ta), tb=bool(tb)).cpu().double()
    assert Fh.gemm_precision.current == Fh._lib.DEFAULT_GEMM_PRECISION
    fast = M % 64 == 0 and N % 128 == 0 and K % 32 == 0
    if fast:
        want = opA.bfloat16().double() @ opB.bfloat16().double()
        assert (got - want).abs().max().item() < 3e-6 * K ** 0.5 * 4 + 1e-6          # only the fp32 accumulation differs
        exact = opA.double() @ opB.double()
        assert (got - exact).abs().max().item() > 1e-5                                # and it really ran in bf16
    else:
        want = opA.double() @ opB.double()
        assert (got - want).abs().max().item() < 2e-6 * K ** 0.5 * 4 + 1e-6
    again = Fh.gemm(A.to(dev), B.to(dev), ta=bool(ta), tb=bool(tb)).cpu().double()
    assert (again - opA.double() @ opB.double()).abs().max().item() < 2e-6 * K ** 0.5 * 4 + 1e-6


@pytest.mark.parametrize("ta,tb,M,N,K,bias", [(0, 1, 4096, 128, 3, 0), (0, 0, 4096, 128, 3, 0), (0, 0, 2048, 256, 16, 0), (0, 1, 4096, 3, 128, 0),
                                              (0, 1, 4096, 16, 256, 1), (0, 0, 4096, 3, 128, 0), (1, 0, 3, 128, 4096, 0),
                                              (1, 0, 16, 256, 8192, 0), (1, 0, 128, 3, 4096, 0), (1, 0, 256, 16, 2048, 0), (1, 0, 64, 3, 8192, 0), (1, 0, 3, 64, 8192, 0),
                                              (0, 1, 32768, 128, 3, 1), (1, 0, 3, 128, 32768, 0)])
def test_thin_gemms(dev, ta, tb, M, N, K, bias):
    """One tiny dimension (3 coordinates, 3 / 16 outputs): the streaming VALU kernels of thin.hip behind mlsp_gemm_f32."""
    Fh = _fh()
    A = _rand((K, M) if ta else (M, K), 21)
    B = _rand((N, K) if tb else (K, N), 22)
    bv = _rand((N,), 23) if bias else None
    want = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
    if bias:
        want = want + bv.double()
    got = Fh.gemm(A.to(dev), B.to(dev), ta=bool(ta), tb=bool(tb), bias=bv.to(dev) if bias else None).cpu().double()
    err = (got - want).abs().max().item()
    assert err < 2e-6 * K ** 0.5 * 4 + 1e-6, err
    again = Fh.gemm(A.to(dev), B.to(dev), ta=bool(ta), tb=bool(tb), bias=bv.to(dev) if bias else None).cpu().double()
    assert torch.equal(got, again)                                     # fixed summation order: bitwise reproducible


def test_gemm_matches_fma_chain_bitwise(dev):
    """The f32 MFMA is a fmaf chain in issue order (what makes the MFMA kNN canonical).  The GEMM kernel feeds K in
    groups of four as (4m, 4m+2, 4m+1, 4m+3) -- see gemm.hip -- so emulate exactly that chain and compare bitwise."""
    Fh = _fh()
    A, B = _rand((64, 24), 3), _rand((40, 24), 4)
    got = Fh.gemm(A.to(dev), B.to(dev), tb=True).cpu().numpy()
    want = np.zeros((64, 40), np.float32)
    a64, b64 = A.numpy().astype(np.float64), B.numpy().astype(np.float64)
    acc = np.zeros((64, 40), np.float64)
    order = [4 * m + d for m in range(6) for d in (0, 2, 1, 3)]
    for kk in order:            # fmaf: exact product + one rounding per step, emulated in float64 (exact for f32 inputs)
        acc = (acc + np.outer(a64[:, kk], b64[:, kk])).astype(np.float32).astype(np.float64)
    want = acc.astype(np.float32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("ta,tb,M,N,K", [(False, True, 32768, 256, 512), (False, False, 32768, 512, 256), (True, False, 512, 256, 32768),
                                          (False, True, 4096, 1024, 128), (False, True, 32768, 1024, 512), (False, False, 65536, 1024, 128),
                                          (True, True, 256, 256, 16384), (True, False, 1024, 512, 32768), (True, False, 2048, 2048, 512),
                                          (True, True, 2048, 2560, 256), (False, False, 2048, 2048, 512)])
@pytest.mark.parametrize("mode,scale", [("bf16x6", 1.0), ("f16x3", 1.0), ("f16x3", 1e-6), ("f16x3", 3e5)])
def test_gemm_split_bf16_accuracy(dev, ta, tb, M, N, K, mode, scale):
    """gemm_precision("bf16x6"): products as six bf16 piece products on the bf16 matrix cores; gemm_precision("f16x3"): as THREE f16 piece
    products (two 11-bit pieces per operand, scaled by a per-workgroup power of two taken from the operands' measured magnitudes -- also
    with the operands at 1e-6 and 3e5 of their size: the scale, not the f16 range, decides).  The bar is the same for both.
    Against float64 its error must be at the
    level of the exact-fp32 MFMA kernel's (both are fp32 accumulations of products exact to <= 2^-25): <= 2x that error and
    <= 2e-6 relative L2 on random operands with a wide dynamic range.  The shapes cover both tile heights (64 / 128 rows), split-K, and
    all four operand layouts (row-major images read with ds_read_b128, k-major images read with ds_read_b64_tr_b16), the last three with a
    k-major / row-major A operand on the 64-row tile (256-511 tiles of 128 rows, no split-K)."""
    Fh = _fh()
    g = torch.Generator().manual_seed(11)
    shpA, shpB = ((K, M) if ta else (M, K)), ((N, K) if tb else (K, N))
    A = (torch.randn(shpA, generator=g) * torch.exp(2.0 * torch.randn(shpA, generator=g))).to(dev) * scale
    B = (torch.randn(shpB, generator=g) * torch.exp(2.0 * torch.randn(shpB, generator=g))).to(dev) * scale
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    with Fh.gemm_precision("fp32"):
        exact = Fh.gemm(A, B, ta, tb).double()
    with Fh.gemm_precision(mode):
        split = Fh.gemm(A, B, ta, tb).double()
    e32 = ((exact - ref).norm() / ref.norm()).item()
    e6 = ((split - ref).norm() / ref.norm()).item()
    m32 = ((exact - ref).abs().max() / ref.abs().max()).item()
    m6 = ((split - ref).abs().max() / ref.abs().max()).item()
    print("rel-L2 vs float64: fp32 MFMA %.2e, %s %.2e | max-abs / max: %.2e, %.2e" % (e32, mode, e6, m32, m6))
    assert not torch.equal(exact, split)                               # it really ran the other kernel
    assert e6 < 2e-6 and e6 < 2.0 * e32 + 1e-8, (e6, e32)
    assert m6 < 2.0 * m32 + 1e-7, (m6, m32)
    assert Fh.gemm_precision.current == Fh._lib.DEFAULT_GEMM_PRECISION


def test_gemm_split_short_k_stays_on_fp32_kernel(dev):
    """K loops of fewer than 4 tiles (8 when N < 256) lose on the split kernel's prologue: mode "bf16x6" keeps them on the f32 MFMA kernel,
    bit for bit."""
    Fh = _fh()
    for M, N, K in ((16384, 128, 64), (4096, 128, 128), (1024, 512, 96)):
        A, B = _rand((M, K), 3).to(dev), _rand((N, K), 4).to(dev)
        with Fh.gemm_precision("fp32"):
            want = Fh.gemm(A, B, False, True)
        with Fh.gemm_precision("bf16x6"):
            got = Fh.gemm(A, B, False, True)
        assert torch.equal(got, want), (M, N, K)


def test_precision_is_per_call_two_threads_and_late_backward(dev):
    """ABI v8: the product mode is an argument of each call, not a library switch.  (1) Two threads that run layers in DIFFERENT modes at
    the same time (own streams, 40 interleaved rounds) get bit for bit what each mode gives alone.  (2) A backward that runs after its
    `with gemm_precision(...)` block has exited uses the forward's products (the mode travels in the autograd context)."""
    import threading
    Fh = _fh()
    M, Cin, Cout = 32768, 256, 256        # (enough tiles that K is not split: the K = 256 loop then runs on the split kernel in mode "bf16x6")
    X = _rand((M, Cin), 31).to(dev)
    W = _rand((Cout, Cin), 32).to(dev)
    gamma, beta = (_rand((Cout,), 33).abs() + 0.5).to(dev), _rand((Cout,), 34).to(dev)

    def layer(mode, stream=None):
        xs, ws = X.clone().requires_grad_(True), W.clone().requires_grad_(True)
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            with Fh.gemm_precision(mode):
                z = Fh.pointmlp(xs, ws, gamma=gamma, beta=beta, training=True, act=Fh.ACT_RELU)
                c = Fh.gemm(X, W, False, True)
            z.square().sum().backward()            # outside the block: the context's mode decides, not the thread's current one
            torch.cuda.current_stream().synchronize()
        return z.detach(), c, xs.grad, ws.grad

    alone = {m: layer(m) for m in ("fp32", "bf16x6")}
    assert not torch.equal(alone["fp32"][0], alone["bf16x6"][0]) and not torch.equal(alone["fp32"][3], alone["bf16x6"][3])
    errs, barrier = [], threading.Barrier(2)

    def worker(mode):
        try:
            st = torch.cuda.Stream()
            for _ in range(40):
                barrier.wait(30)
                got = layer(mode, st)
                for a, b in zip(got, alone[mode]):
                    assert torch.equal(a, b), mode
        except Exception as e:                      # noqa: BLE001
            errs.append((mode, repr(e)))
            barrier.abort()
    ts = [threading.Thread(target=worker, args=(m,)) for m in ("fp32", "bf16x6")]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert Fh.gemm_precision.current == Fh._lib.DEFAULT_GEMM_PRECISION


# ----------------------------------------------------------------------------- bf16 activation storage (configs[4])
class _RoundBF16(torch.autograd.Function):
    """x -> bf16(x) with a straight-through gradient: where the storage-mode kernels round, the emulation rounds."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


@pytest.mark.parametrize("M,C0,training", [(32768, 192, True), (32768, 512, True), (32768, 192, False)])
def test_pointmlp_bf16_activation_storage(dev, M, C0, training):
    """functional.activation_storage("bf16"): a chain of Linear+BN+act layers keeps Y / Z (and dZ / dY / dX in backward) as bf16 in HBM
    (mlsp_pointmlp_*_mx: bf16 x bf16 products, fp32 accumulation, fp32 BN statistics from the accumulators).
    Oracle: the same chain in plain torch fp32 with a bf16 rounding (straight-through in backward) at every point where the kernels
    round -- GEMM operands, stored Y, stored Z.  (Against the un-rounded fp32 chain the OUTPUT agrees to ~1 %, but its gradient
    differs by ~9 %: rounding Y flips the ReLU mask of the ~0.3 % of activations that sit at the kink, i.e. it is the exact gradient
    of a slightly different function.)"""
    Fh = _fh()
    dims = [C0, 256, 256, 128]
    g = torch.Generator().manual_seed(5)
    X = torch.randn(M, C0, generator=g)
    Ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(3)]
    gam = [torch.rand(dims[i + 1], generator=g) + 0.5 for i in range(3)]
    bet = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(3)]
    rms = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(3)]
    rvs = [torch.rand(dims[i + 1], generator=g) + 0.5 for i in range(3)]
    gb = torch.randn(M // 1024, 256, generator=g) * 0.3
    wout = torch.randn(M, 128, generator=g)
    rnd = _RoundBF16.apply

    def leaves():
        return ([X.to(dev).requires_grad_(True)] + [w.to(dev).requires_grad_(True) for w in Ws] + [t.to(dev).requires_grad_(True) for t in gam] +
                [t.to(dev).requires_grad_(True) for t in bet] + [gb.to(dev).requires_grad_(True)])

    def run(mode):
        L = leaves()
        x, ws, gs, bs, gbd = L[0], L[1:4], L[4:7], L[7:10], L[10]
        seen = []
        with Fh.activation_storage(mode):
            h = x
            for i in range(3):
                h = Fh.pointmlp(h, ws[i], gbias=gbd if i == 0 else None, rows_per_group=1024 if i == 0 else 0, gamma=gs[i], beta=bs[i],
                                run_mean=rms[i].to(dev), run_var=rvs[i].to(dev), training=training, act=Fh.ACT_RELU, chain=(i < 2))
                seen.append(h.dtype)
        (h.float() * wout.to(dev)).sum().backward()
        return h.detach().float().cpu(), [t.grad.float().cpu() for t in L], seen

    def emulate():
        L = leaves()
        x, ws, gs, bs, gbd = L[0], L[1:4], L[4:7], L[7:10], L[10]
        h = x
        for i in range(3):
            store = i < 2                                              # layers 0, 1 store bf16; layer 2 reads bf16, writes fp32
            y = rnd(h) @ rnd(ws[i]).t()                                # operands rounded, fp32 accumulation
            if i == 0:
                y = y + gbd.repeat_interleave(1024, dim=0)
            if training:
                mean, var = y.mean(0), y.var(0, unbiased=False)        # statistics from the fp32 accumulators
            else:
                mean, var = rms[i].to(dev), rvs[i].to(dev)
            scale = gs[i] * torch.rsqrt(var + 1e-5)
            yb = rnd(y) if store else y
            z = torch.relu(yb * scale + (bs[i] - mean * scale))
            h = rnd(z) if store else z
        (h * wout.to(dev)).sum().backward()
        return h.detach().cpu(), [t.grad.cpu() for t in L]

    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-30)).item()
    ref32, _, dref = run("fp32")
    got, ggot, dgot = run("bf16")
    want, gwant = emulate()
    assert dref == [torch.float32] * 3 and dgot == [torch.bfloat16, torch.bfloat16, torch.float32]
    assert Fh.activation_storage.current == "fp32"
    assert rel(got, want) < 5e-3, rel(got, want)                       # same roundings at the same places
    assert 1e-5 < rel(got, ref32) < 2e-2, rel(got, ref32)              # close to the fp32 chain, and it really stored bf16
    errs = [rel(a, b) for a, b in zip(ggot, gwant)]
    print("bf16-storage gradients vs rounded-forward emulation:", ["%.1e" % e for e in errs])
    assert max(errs) < 3e-2, errs                                      # gradient tensors themselves are stored / multiplied in bf16


@pytest.mark.parametrize("M", [8192, 16384, 32768])
def test_bf16_storage_producer_consumer_shapes_disagree(dev, M):
    """under activation_storage("bf16") a chained producer decides its output type from its OWN shape; at M = 8192 / 16384
    the consumer's GEMM splits K and is outside the bf16-storage kernels.  The stack must run (the consumer widens its input once)
    and agree with the fp32-storage stack to bf16 accuracy, forward and backward."""
    Fh = _fh()
    dims = [(512, 256), (256, 256), (256, 128)]
    X = _rand((M, 512), 1).to(dev)
    Ws = [(_rand((co, ci), 10 + i, 1.0 / ci ** 0.5)).to(dev) for i, (ci, co) in enumerate(dims)]
    gs = [(1.0 + 0.1 * _rand((co,), 20 + i)).to(dev) for i, (_, co) in enumerate(dims)]
    bs = [(0.1 * _rand((co,), 30 + i)).to(dev) for i, (_, co) in enumerate(dims)]

    def run(store, prec):
        x = X.clone().requires_grad_(True)
        ws = [w.clone().requires_grad_(True) for w in Ws]
        with Fh.gemm_precision(prec), Fh.activation_storage(store):
            h = x
            for i in range(3):
                h = Fh.pointmlp(h, ws[i], gamma=gs[i], beta=bs[i], training=True, act=Fh.ACT_RELU, chain=(i < 2))
            h.float().square().mean().backward()
        return h.float().detach(), x.grad, [w.grad for w in ws]

    o32, gx32, gw32 = run("fp32", "fp32")
    o16, gx16, gw16 = run("bf16", "bf16")

    def rel(a, b):
        return ((a - b).norm() / (b.norm() + 1e-12)).item()
    assert rel(o16, o32) < 3e-2, rel(o16, o32)
    assert rel(gx16, gx32) < 0.15, rel(gx16, gx32)       # (a rounded Y flips ReLU masks at the kink: see the emulation test above)
    for a, b in zip(gw16, gw32):
        assert rel(a, b) < 0.15, rel(a, b)


# ----------------------------------------------------------------------------- pointmlp (Linear + BN + act)
def _torch_pointmlp(X, W, bias, gbias, rpg, gamma, beta, rm, rv, training, act):
    Y = X @ W.t()
    if bias is not None:
        Y = Y + bias
    if gbias is not None:
        Y = Y + gbias.repeat_interleave(rpg, dim=0)
    if gamma is not None:
        Y = F.batch_norm(Y, rm, rv, gamma, beta, training, 0.1, 1e-5)
    if act == 1:
        Y = F.relu(Y)
    elif act == 2:
        Y = F.leaky_relu(Y, 0.2)
    return Y


@pytest.mark.parametrize("M,Cin,Cout,act,use_bias,G,training", [
    (1000, 64, 128, 2, False, 0, True), (512, 6, 64, 2, False, 0, True), (32, 1024, 512, 2, True, 0, True),
    (2048, 512, 256, 1, False, 4, True), (777, 100, 33, 1, True, 0, False), (640, 128, 3, 0, True, 0, True),
    # per-cloud bias gradient out of the BN-backward pass (1024-row clouds: 16 slabs) and its separate-kernel fallback (ragged clouds)
    (2048, 128, 256, 2, False, 2, True), (1200, 64, 128, 2, False, 4, True),
    # 64 output channels with a bias and fused statistics on the 128 x 64-tile kernel (needs >= 1536 row panels)
    (196608, 64, 64, 1, True, 0, True),
    # per-cloud layers (skinny.hip): fused Linear + BN1d + act over <= 32 rows; eval mode; ragged sizes; plain Linear
    (32, 512, 256, 2, False, 0, True), (20, 300, 70, 1, True, 0, False), (5, 64, 40, 2, True, 0, True),
    (32, 256, 9, 0, True, 0, True), (32, 256, 10, 0, True, 0, False)])
def test_pointmlp_fwd_bwd(dev, M, Cin, Cout, act, use_bias, G, training):
    Fh = _fh()
    has_bn = act != 0
    X = _rand((M, Cin), 1).requires_grad_(True)
    W = _rand((Cout, Cin), 2, 0.2).requires_grad_(True)
    bias = _rand((Cout,), 3).requires_grad_(True) if use_bias else None
    rpg = M // G if G else 0
    gbias = _rand((G, Cout), 4).requires_grad_(True) if G else None
    gamma = (_rand((Cout,), 5) + 0.2).requires_grad_(True) if has_bn else None      # includes negative scales
    beta = _rand((Cout,), 6).requires_grad_(True) if has_bn else None
    rm, rv = _rand((Cout,), 7) * 0.1, _rand((Cout,), 8).abs() + 0.5
    dZ = _rand((M, Cout), 9)

    rm_c, rv_c = rm.clone(), rv.clone()
    Zc = _torch_pointmlp(X, W, bias, gbias, rpg, gamma, beta, rm_c if has_bn else None, rv_c if has_bn else None, training, act)
    Zc.backward(dZ)
    leaves = [t for t in (X, W, bias, gbias, gamma, beta) if t is not None]
    want = [t.grad.clone() for t in leaves]

    def d(t):
        return t.detach().to(dev).requires_grad_(True) if t is not None else None
    Xg, Wg, bg, gbg, gg, betag = d(X), d(W), d(bias), d(gbias), d(gamma), d(beta)
    rm_g, rv_g = rm.to(dev), rv.to(dev)
    Zg = Fh.pointmlp(Xg, Wg, bias=bg, gbias=gbg, gamma=gg, beta=betag, run_mean=rm_g if has_bn else None,
                     run_var=rv_g if has_bn else None, rows_per_group=rpg, training=training, act=act)
    Zg.backward(dZ.to(dev))
    np.testing.assert_allclose(Zg.detach().cpu().numpy(), Zc.detach().numpy(), rtol=1e-4, atol=2e-4)
    got = [t.grad for t in (Xg, Wg, bg, gbg, gg, betag) if t is not None]
    names = [n for n, t in zip(["dX", "dW", "dbias", "dgbias", "dgamma", "dbeta"], (X, W, bias, gbias, gamma, beta))
             if t is not None]
    for gw, ww, name in zip(got, want, names):
        scale = ww.abs().max().item() + 1e-6
        err = (gw.cpu() - ww).abs().max().item()
        if name == "dbias" and has_bn and training:
            assert err < 1e-3 * (dZ.abs().sum(0).max().item()), (name, err)      # analytically zero
        else:
            assert err / scale < 2e-3, (name, err, scale)
    if has_bn and training:
        np.testing.assert_allclose(rm_g.cpu().numpy(), rm_c.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rv_g.cpu().numpy(), rv_c.numpy(), rtol=1e-4, atol=1e-5)


def test_pointmlp_weight_column_slice(dev):
    """W passed as a column slice of a wider weight (the heads' 1536-channel conv1 split)."""
    Fh = _fh()
    X = _rand((256, 40), 1)
    Wfull = _rand((16, 100), 2, 0.3)
    Wg = Wfull.to(dev).requires_grad_(True)
    Z = Fh.pointmlp(X.to(dev), Wg[:, :40])
    Z.sum().backward()
    np.testing.assert_allclose(Z.detach().cpu().numpy(), (X @ Wfull[:, :40].t()).numpy(), rtol=1e-4, atol=1e-4)
    want = torch.zeros_like(Wfull)
    want[:, :40] = X.sum(0, keepdim=True).expand(16, 40)
    np.testing.assert_allclose(Wg.grad.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-3)


def test_dropout_statistics_and_mask_consistency(dev):
    Fh = _fh()
    M, C = 4096, 128
    X = torch.ones(M, C)
    W = torch.eye(C)
    gamma, beta = torch.ones(C), torch.full((C,), 2.0)
    Xg = X.to(dev).requires_grad_(True)
    torch.manual_seed(7)
    Z = Fh.pointmlp(Xg + 0.001 * _rand((M, C), 3).to(dev), W.to(dev), gamma=gamma.to(dev), beta=beta.to(dev),
                    run_mean=torch.zeros(C, device=dev), run_var=torch.ones(C, device=dev), training=True, act=1, p_drop=0.5)
    keep = (Z != 0)
    frac = keep.float().mean().item()
    assert abs(frac - 0.5) < 0.01, frac
    # kept values are scaled by 1/(1-p)
    assert (Z[keep] > 0).all()
    col_frac = keep.float().mean(0)
    assert (col_frac - 0.5).abs().max().item() < 0.06
    Z.backward(torch.ones_like(Z))
    # eval: no dropout
    Ze = Fh.pointmlp(X.to(dev), W.to(dev), gamma=gamma.to(dev), beta=beta.to(dev), run_mean=torch.zeros(C, device=dev),
                     run_var=torch.ones(C, device=dev), training=False, act=1, p_drop=0.5)
    assert (Ze != 0).all()


def test_skinny_dropout_mask_consistent_fwd_bwd(dev):
    """Per-cloud FC layer (32 rows, skinny.hip) with dropout: the backward must regenerate the forward's keep mask."""
    Fh = _fh()
    M, Cin, C = 32, 64, 256
    X, W = _rand((M, Cin), 1).requires_grad_(True), _rand((C, Cin), 2, 0.3).requires_grad_(True)
    gamma, beta = (_rand((C,), 3) + 0.3).requires_grad_(True), _rand((C,), 4).requires_grad_(True)
    dZ = _rand((M, C), 5)
    Xg, Wg, gg, bg = (t.detach().to(dev).requires_grad_(True) for t in (X, W, gamma, beta))
    Z = Fh.pointmlp(Xg, Wg, gamma=gg, beta=bg, run_mean=torch.zeros(C, device=dev), run_var=torch.ones(C, device=dev),
                    training=True, act=2, p_drop=0.5)
    Z.backward(dZ.to(dev))
    mask = (Z.detach().cpu() != 0).float()
    assert abs(mask.mean().item() - 0.5) < 0.03
    A = F.leaky_relu(F.batch_norm(X @ W.t(), None, None, gamma, beta, True, 0.1, 1e-5), 0.2)
    Zc = A * mask * 2.0
    Zc.backward(dZ)
    np.testing.assert_allclose(Z.detach().cpu().numpy(), Zc.detach().numpy(), rtol=1e-4, atol=2e-4)
    for got, want, name in ((Xg.grad, X.grad, "dX"), (Wg.grad, W.grad, "dW"), (gg.grad, gamma.grad, "dgamma"), (bg.grad, beta.grad, "dbeta")):
        err = (got.cpu() - want).abs().max().item() / (want.abs().max().item() + 1e-6)
        assert err < 2e-3, (name, err)


# ----------------------------------------------------------------------------- several Linear layers under one BatchNorm pass
@pytest.mark.parametrize("M,training", [(32768, True), (2048, True), (4096, False)])
def test_multimlp_vs_one_pointmlp_per_segment(dev, M, training):
    """csrc/multi.hip: three Linear layers side by side (two 256->256 without bias, one 512->256 with bias; ReLU / ReLU / LeakyReLU)
    under ONE BatchNorm + activation pass against one Fh.pointmlp per segment: outputs, input gradient, weight / bias / BatchNorm
    gradients and running statistics.  M = 32768: statistics out of the GEMM epilogues; M = 2048: split-K segments."""
    Fh = _fh()
    g = torch.Generator().manual_seed(3)
    X = torch.randn(M, 1024, generator=g).to(dev)
    specs = [(0, 256, 256, False, Fh.ACT_RELU, 0.0), (256, 256, 256, False, Fh.ACT_RELU, 0.0), (512, 512, 256, True, Fh.ACT_LRELU, 0.2)]
    Ws = [(torch.randn(co, ci, generator=g) / ci ** 0.5).to(dev) for _, ci, co, _, _, _ in specs]
    bs = [(0.1 * torch.randn(co, generator=g)).to(dev) if hb else None for _, _, co, hb, _, _ in specs]
    gam = [(torch.rand(co, generator=g) + 0.5).to(dev) * (1 if i else -1) for i, (_, _, co, _, _, _) in enumerate(specs)]
    bet = [(0.1 * torch.randn(co, generator=g)).to(dev) for _, _, co, _, _, _ in specs]
    wout = torch.randn(M, 768, generator=g).to(dev)

    def run(multi):
        x = X.clone().requires_grad_(True)
        ws = [w.clone().requires_grad_(True) for w in Ws]
        bb = [b.clone().requires_grad_(True) if b is not None else None for b in bs]
        gs = [t.clone().requires_grad_(True) for t in gam]
        be = [t.clone().requires_grad_(True) for t in bet]
        rms = [torch.zeros(256, device=dev) for _ in specs]
        rvs = [torch.ones(256, device=dev) for _ in specs]
        if multi:
            chan = Fh.channel_params(dev, tuple((co, sl, False) for _, _, co, _, _, sl in specs))
            rm, rv = Fh.merged_buffers(rms, rehome=False), Fh.merged_buffers(rvs, rehome=False)
            assert Fh.multimlp_supported(M, x, ws, [sp[0] for sp in specs])
            z = Fh.multimlp(x, [(sp[0], w, b) for sp, w, b in zip(specs, ws, bb)], torch.cat(gs), torch.cat(be), rm.tensor, rv.tensor, chan,
                            training=training)
            rm.writeback(); rv.writeback()
        else:
            z = torch.cat([Fh.pointmlp(x[:, xc:xc + ci], w, bias=b, gamma=ga, beta=bt, run_mean=r1, run_var=r2, training=training, act=act)
                           for (xc, ci, co, hb, act, sl), w, b, ga, bt, r1, r2 in zip(specs, ws, bb, gs, be, rms, rvs)], dim=1)
        (z * wout).sum().backward()
        grads = [x.grad] + [w.grad for w in ws] + [b.grad for b in bb if b is not None] + [t.grad for t in gs] + [t.grad for t in be]
        return z.detach(), grads, rms + rvs

    za, ga_, sa = run(False)
    zb, gb_, sb = run(True)
    np.testing.assert_allclose(zb.cpu().numpy(), za.cpu().numpy(), rtol=1e-4, atol=1e-5)
    for a, b in zip(ga_, gb_):
        assert ((a - b).norm() / (a.norm() + 1e-20)).item() < 1e-4
    for a, b in zip(sa, sb):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_multimlp_dropout_per_channel(dev):
    """dropout switched per channel: the first segment's channels drop with rate 0.5 (zeros, survivors scaled by 2), the second's never;
    backward uses the same mask (zero input gradient contribution exactly where the output was dropped)."""
    Fh = _fh()
    M = 4096
    g = torch.Generator().manual_seed(5)
    X = torch.randn(M, 256, generator=g).to(dev).requires_grad_(True)
    W1, W2 = torch.eye(128).to(dev).requires_grad_(True), torch.eye(128).to(dev).requires_grad_(True)
    gamma, beta = torch.ones(256, device=dev), torch.full((256,), 3.0, device=dev)          # outputs positive: no activation zeros
    chan = Fh.channel_params(dev, ((128, 1.0, True), (128, 1.0, False)))
    z = Fh.multimlp(X, [(0, W1, None), (128, W2, None)], gamma, beta, None, None, chan, training=True, p_drop=0.5)
    dropped = (z[:, :128] == 0)
    frac = dropped.float().mean().item()
    assert 0.48 < frac < 0.52 and not (z[:, 128:] == 0).any().item()
    # survivors are 2 x the un-dropped value
    z0 = Fh.multimlp(X.detach(), [(0, W1.detach(), None), (128, W2.detach(), None)], gamma, beta, None, None, chan, training=True, p_drop=0.0)
    np.testing.assert_allclose(z[:, :128][~dropped].detach().cpu().numpy(), 2 * z0[:, :128][~dropped].cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(z[:, 128:].detach().cpu().numpy(), z0[:, 128:].cpu().numpy(), rtol=1e-6)
    torch.manual_seed(0)
    wout = torch.randn(M, 256, device=dev)
    (z * wout).sum().backward()
    assert torch.isfinite(X.grad).all().item() and torch.isfinite(W1.grad).all().item()


# ----------------------------------------------------------------------------- graph feature / max reductions
def test_graph_feature_fwd_bwd(dev, golden_dir):
    from mlsp_amd import model_utils as mu
    g = dict(np.load(os.path.join(golden_dir, "graph_feature.npz")))
    x = torch.from_numpy(g["x"])
    xg = x.to(dev).requires_grad_(True)
    f = mu.get_graph_feature(xg, gc.make_args(cuda=True), k=20)
    assert f.shape == g["feat"].shape
    np.testing.assert_allclose(f.detach().cpu().numpy(), g["feat"], rtol=0, atol=0)
    w = _rand(tuple(f.shape), 3)
    (f * w.to(dev)).sum().backward()
    xc = x.clone().requires_grad_(True)
    fc = ref_cpu.graph_feature(xc, ref_cpu.knn_reference_formula(xc, 20))
    (fc * w).sum().backward()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-4, atol=1e-4)
    # caller-provided indices
    idx = ref_cpu.knn_reference_formula(x, 20)
    f2 = mu.get_graph_feature(x.to(dev), gc.make_args(cuda=True), k=20, idx=idx.to(dev))
    np.testing.assert_allclose(f2.cpu().numpy(), g["feat"], rtol=0, atol=0)


def test_segmax_colmax(dev):
    Fh = _fh()
    Z = _rand((50 * 20, 70), 1)
    Zg = Z.to(dev).requires_grad_(True)
    out = Fh.segmax(Zg, 20)
    w = _rand((50, 70), 2)
    (out * w.to(dev)).sum().backward()
    Zc = Z.clone().requires_grad_(True)
    oc = Zc.view(50, 20, 70).max(1)[0]
    (oc * w).sum().backward()
    assert torch.equal(out.detach().cpu(), oc.detach())
    assert torch.equal(Zg.grad.cpu(), Zc.grad)
    Y = _rand((3 * 100, 130), 3)
    Yg = Y.to(dev).requires_grad_(True)
    o2 = Fh.colmax(Yg, 3, 100)
    w2 = _rand((3, 130), 4)
    (o2 * w2.to(dev)).sum().backward()
    Yc = Y.clone().requires_grad_(True)
    o2c = Yc.view(3, 100, 130).max(1)[0]
    (o2c * w2).sum().backward()
    assert torch.equal(o2.detach().cpu(), o2c.detach())
    assert torch.equal(Yg.grad.cpu(), Yc.grad)


# ----------------------------------------------------------------------------- fused EdgeConv
def _torch_edgeconv(xp, idx, W, gamma, beta, rm, rv, training, B, N):
    """graph feature -> 1x1 conv -> BN2d -> LeakyReLU -> max over k, the reference's way, on [P,C]."""
    C = xp.shape[1]
    x = xp.view(B, N, C).transpose(2, 1)
    f = ref_cpu.graph_feature(x, idx)                                  # [B,2C,N,k]
    y = torch.einsum("oc,bcnk->bonk", W, f)
    y = F.batch_norm(y, rm, rv, gamma, beta, training, 0.1, 1e-5)
    y = F.leaky_relu(y, 0.2).max(dim=-1)[0]                            # [B,Cout,N]
    return y.transpose(2, 1).reshape(B * N, -1)


@pytest.mark.parametrize("B,N,C,Cout,training,k", [(2, 128, 3, 64, True, 20), (2, 96, 64, 64, True, 20), (3, 64, 64, 128, True, 20),
                                                   (1, 200, 128, 256, True, 20), (2, 100, 16, 40, False, 20),
                                                   # k = 40 (BASELINE.json configs[4]) / 36 / 28: the LDS-resident gather-reduce with 40- and 32-slot lists
                                                   (2, 128, 3, 64, True, 40), (2, 256, 64, 64, True, 40), (1, 160, 64, 128, True, 36),
                                                   (2, 128, 64, 64, False, 28)])
def test_edgeconv_fwd_bwd(dev, B, N, C, Cout, training, k):
    Fh = _fh()
    P = B * N
    xp = _rand((P, C), 1).requires_grad_(True)
    W = _rand((Cout, 2 * C), 2, 0.3).requires_grad_(True)
    gamma = (_rand((Cout,), 3) + 0.3).requires_grad_(True)             # ~1/3 negative -> min branch
    beta = _rand((Cout,), 4).requires_grad_(True)
    rm, rv = _rand((Cout,), 5) * 0.1, _rand((Cout,), 6).abs() + 0.5
    dOut = _rand((P, Cout), 7)
    idx = torch.from_numpy(knn_canon.knn_point_major(xp.detach().view(B, N, C), k).astype(np.int64))

    rm_c, rv_c = rm.clone(), rv.clone()
    oc = _torch_edgeconv(xp, idx, W, gamma, beta, rm_c, rv_c, training, B, N)
    oc.backward(dOut)

    xg, Wg, gg, bg = [t.detach().to(dev).requires_grad_(True) for t in (xp, W, gamma, beta)]
    rm_g, rv_g = rm.to(dev), rv.to(dev)
    graph = Fh.knn_graph(xg, B, N, k)
    assert np.array_equal(graph.idx.view(B, N, k).cpu().numpy(), idx.numpy())
    og = Fh.edgeconv(xg, graph, Wg, gg, bg, rm_g, rv_g, training)
    og.backward(dOut.to(dev))
    np.testing.assert_allclose(og.detach().cpu().numpy(), oc.detach().numpy(), rtol=2e-4, atol=2e-4)
    for got, want, name in [(xg.grad, xp.grad, "dx"), (Wg.grad, W.grad, "dW"), (gg.grad, gamma.grad, "dgamma"),
                            (bg.grad, beta.grad, "dbeta")]:
        scale = want.abs().max().item() + 1e-6
        err = (got.cpu() - want).abs().max().item()
        assert err / scale < 2e-3, (name, err, scale)
    if training:
        np.testing.assert_allclose(rm_g.cpu().numpy(), rm_c.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rv_g.cpu().numpy(), rv_c.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("gscale", [1e-6, 1.0, 3e4])
def test_edgeconv_backward_f16x3_follows_the_gradient_magnitude(dev, gscale):
    """Mode "f16x3" at the headline's conv4 (P = 32768, 128 -> 256, k = 20): the passes that write the point-space gradient duv leave its
    bound as a by-product (edge.hip edge_amax_raise) and the two products that read it scale their pieces by it.  A bound that is too
    small overflows the f16 pieces, one that is far too large costs significand bits: both show against a float64 evaluation of the
    layer (torch ops on the GPU), at three magnitudes of the incoming gradient.  Bar: the error of the f32-MFMA mode of the same call
    (x 2, as test_gemm_split_bf16_accuracy), on the output and on the median row of dx.  The products must really run on the two-piece kernel (profile hook).
    Figures: profiles/r6_edge_bwd_probe.txt (tools/r6/edge_bwd_probe.py)."""
    import ctypes
    from mlsp_amd import _lib
    Fh = _fh()
    B, N, C, Cout, k = 32, 1024, 128, 256, 20
    P = B * N
    g = torch.Generator().manual_seed(5)
    xp = torch.randn(P, C, generator=g).to(dev)
    W = (torch.randn(Cout, 2 * C, generator=g) * 0.1).to(dev)
    gamma, beta = (torch.randn(Cout, generator=g) + 0.3).to(dev), torch.randn(Cout, generator=g).to(dev)
    dOut = (torch.randn(P, Cout, generator=g) * gscale).to(dev)
    graph = Fh.knn_graph(xp, B, N, k)
    idx = graph.idx.view(B, N, k).long()
    # float64: get_graph_feature -> 1x1 conv -> BatchNorm2d (batch statistics) -> LeakyReLU -> max over k (PointDA/Models.py:115-129)
    x64, W64 = xp.double().requires_grad_(True), W.double().requires_grad_(True)
    xb = x64.view(B, N, C)
    nb = torch.gather(xb.unsqueeze(1).expand(B, N, N, C), 2, idx.unsqueeze(-1).expand(B, N, k, C))
    y = torch.cat((nb - xb.unsqueeze(2), xb.unsqueeze(2).expand(B, N, k, C)), dim=-1) @ W64.t()
    y = (y - y.mean(dim=(0, 1, 2))) / torch.sqrt(y.var(dim=(0, 1, 2), unbiased=False) + 1e-5) * gamma.double() + beta.double()
    o64 = F.leaky_relu(y, 0.2).max(dim=2)[0].reshape(P, Cout)
    o64.backward(dOut.double())
    want = (o64.detach(), x64.grad, W64.grad)
    del y, nb
    err, lib = {}, _lib.load()
    for mode in ("fp32", "f16x3"):
        xg, Wg, gg, bg = [t.clone().requires_grad_(True) for t in (xp, W, gamma, beta)]
        rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        with Fh.gemm_precision(mode):
            lib.mlsp_profile_begin()
            out = Fh.edgeconv(xg, graph, Wg, gg, bg, rm, rv, True)
            out.backward(dOut)
            torch.cuda.synchronize()
            buf, kinds = (ctypes.c_double * 4)(), (ctypes.c_double * 16)()
            lib.mlsp_profile_end(buf)
            lib.mlsp_profile_split_kinds(kinds)
        assert int(kinds[13]) == (3 if mode == "f16x3" else 0), (mode, list(kinds))        # forward, dgrad, wgrad on f16 pieces
        got = (out.detach(), xg.grad, Wg.grad)
        assert all(torch.isfinite(t).all() for t in got), mode
        rows = ((got[1].double() - want[1]).norm(dim=1) / want[1].norm(dim=1)).median().item()
        err[mode] = [((a.double() - b).norm() / b.norm()).item() for a, b in zip(got, want)] + [rows]
    # A max over k that flips between two nearly equal candidates moves two rows of dx by O(1) of their size (and dW by ~1e-5 of its
    # norm) in ANY mode -- tools/r6/edge_bwd_probe.py shows 1e-5 .. 7e-4 in the f32-MFMA, bf16x6 and f16x3 modes alike, depending on the
    # data --, so dx is held by its MEDIAN row (a loose bound costs bits in every row, an overflow is not finite), dW loosely.
    assert err["f16x3"][0] <= 2 * err["fp32"][0] and err["f16x3"][0] < 1e-6, (gscale, err)
    assert err["f16x3"][3] <= 2 * err["fp32"][3] and err["f16x3"][3] < 2e-6, (gscale, err)
    assert err["f16x3"][2] < 2e-3, (gscale, err)


# ----------------------------------------------------------------------------- losses
@pytest.mark.parametrize("fname", ["loss_s0_N256.npz", "loss_s1_N1024.npz"])
def test_losses_vs_golden(dev, golden_dir, fname):
    from mlsp_amd import mlsp
    Fh = _fh()
    g = dict(np.load(os.path.join(golden_dir, fname)))
    seed = int(fname.split("_s")[1][0])
    N = int(fname.split("_N")[1].split(".")[0])
    args = gc.make_args()
    inp = {k: v.to(dev) for k, v in gc.make_inputs(seed, 2, N).items()}
    pred = torch.from_numpy(g["pred"]).to(dev).requires_grad_(True)
    normal = torch.from_numpy(g["normal"]).to(dev).requires_grad_(True)
    lg = torch.from_numpy(g["dlogits"]).to(dev).requires_grad_(True)
    fc2w = (torch.arange(16, dtype=torch.float32) * 2.0).view(1, 16).to(dev)
    p, dens = Fh.density_tail(lg, fc2w)
    logits = {"DefRec": pred, "Normal": normal, "density": p, "density_mse": dens}
    loss_def = mlsp.calc_loss(args, logits, inp["gold"], inp["mask"])
    mask_cord = inp["mask"].permute(0, 2, 1)[:, :, 0] * 26 + 1
    nl = mlsp.calc_masked_normal_loss(args, normal, inp["normal_gt"], mask_cord)
    kl, mae = mlsp.densityloss(args, logits, inp["dens_val"], inp["dens_vec"], mask=mask_cord.reshape(-1))
    loss = loss_def + nl + kl + mae
    loss.backward()
    np.testing.assert_allclose(loss_def.item(), g["loss_DefRec"], rtol=1e-4)
    np.testing.assert_allclose(nl.item(), g["loss_normal"], rtol=1e-4)
    np.testing.assert_allclose(kl.item(), g["loss_kl"], rtol=1e-4)
    np.testing.assert_allclose(mae.item(), g["loss_mae"], rtol=1e-4)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    np.testing.assert_allclose(pred.grad.cpu().numpy(), g["g_pred"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(normal.grad.cpu().numpy(), g["g_normal"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), g["g_dlogits"], rtol=1e-3, atol=1e-7)
    with torch.no_grad():
        np.testing.assert_allclose(mlsp.calc_normal_loss(args, normal, inp["normal_gt"]).item(), g["normal_unmasked"], rtol=1e-4)
        kl_u, mae_u = mlsp.densityloss(args, logits, inp["dens_val"], inp["dens_vec"])
        np.testing.assert_allclose(kl_u.item(), g["kl_unmasked"], rtol=1e-4)
        np.testing.assert_allclose(mae_u.item(), g["mae_unmasked"], rtol=1e-4)


@pytest.mark.parametrize("fname", ["chamfer_dir_s4_B3_N256.npz", "chamfer_dir_s5_B2_N1024.npz"])
def test_chamfer_distance_one_direction_vs_golden(dev, golden_dir, fname):
    """mlsp.chamfer_distance(p1, p2, mask) (MLSP/mlsp.py:115-153) against the reference's value and BOTH gradients; the two
    directions add up to reconstruction_loss."""
    from mlsp_amd import mlsp
    g = dict(np.load(os.path.join(golden_dir, fname)))
    p1 = torch.from_numpy(g["p1"]).to(dev).requires_grad_(True)
    p2 = torch.from_numpy(g["p2"]).to(dev).requires_grad_(True)
    mask = torch.from_numpy(g["mask"]).to(dev)
    d = mlsp.chamfer_distance(p1, p2, mask)
    d.backward()
    np.testing.assert_allclose(d.item(), g["dist"], rtol=1e-5)
    np.testing.assert_allclose(p1.grad.cpu().numpy(), g["g_p1"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(p2.grad.cpu().numpy(), g["g_p2"], rtol=1e-4, atol=1e-7)
    with torch.no_grad():
        both = mlsp.chamfer_distance(p1, p2, mask) + mlsp.chamfer_distance(p2, p1, mask)
        fused = mlsp.reconstruction_loss(p2, p1.permute(0, 2, 1).contiguous(), mask.permute(0, 2, 1).contiguous())
    np.testing.assert_allclose(both.item() / p1.shape[0], fused.item(), rtol=1e-6)
    # only one side needs a gradient
    q = torch.from_numpy(g["p2"]).to(dev).requires_grad_(True)
    mlsp.chamfer_distance(p1.detach(), q, mask).backward()
    np.testing.assert_allclose(q.grad.cpu().numpy(), g["g_p2"], rtol=1e-4, atol=1e-7)


def test_chamfer_edge_cases(dev):
    """Empty mask -> NaN like the reference (0/0, mlsp.py:152); all-masked cloud; pred == gold -> 0."""
    from mlsp_amd import mlsp
    B, N = 2, 64
    gold = _rand((B, 3, N), 1)
    pred = gold.permute(0, 2, 1).contiguous()
    mask = torch.zeros(B, 3, N)
    mask[0, :, :10] = 1
    mask[1] = 1
    l = mlsp.reconstruction_loss(pred.to(dev), gold.to(dev), mask.to(dev))
    assert l.item() == 0.0
    want = ref_cpu.reconstruction_loss(pred + 0.1, gold, mask)
    got = mlsp.reconstruction_loss((pred + 0.1).to(dev), gold.to(dev), mask.to(dev))
    np.testing.assert_allclose(got.item(), want.item(), rtol=1e-5)
    mask[0] = 0
    assert torch.isnan(mlsp.reconstruction_loss(pred.to(dev), gold.to(dev), mask.to(dev))).item()


# ----------------------------------------------------------------------------- fused T-Net per-edge stage
@pytest.mark.parametrize("B,N,k,training", [(2, 128, 20, True), (3, 50, 20, True), (1, 77, 7, True), (2, 64, 20, False),
                                            (2, 96, 24, True), (2, 80, 32, True), (2, 100, 40, True), (1, 130, 48, True),
                                            (8, 70, 16, True), (2, 90, 8, True), (3, 66, 10, True), (1, 140, 64, True),
                                            (16, 33, 12, True)])
def test_tnet_edge_fused_vs_materialised(dev, B, N, k, training):
    """tnet.hip (LDS-resident gather + MFMA) against the reference's op sequence in torch on the CPU:
    graph feature -> conv 6->64 + BN + LReLU -> conv 64->128 + BN + LReLU -> max over k."""
    Fh = _fh()
    P = B * N
    xp = _rand((P, 3), 1).requires_grad_(True)
    W1 = _rand((64, 6), 2, 0.5).requires_grad_(True)
    W2 = _rand((128, 64), 3, 0.2).requires_grad_(True)
    g1 = (_rand((64,), 4) + 0.3).requires_grad_(True)
    b1 = _rand((64,), 5).requires_grad_(True)
    g2 = (_rand((128,), 6) + 0.3).requires_grad_(True)
    b2 = _rand((128,), 7).requires_grad_(True)
    rms = [_rand((64,), 8) * 0.1, _rand((64,), 9).abs() + 0.5, _rand((128,), 10) * 0.1, _rand((128,), 11).abs() + 0.5]
    dOut = _rand((P, 128), 12)
    idx = torch.from_numpy(knn_canon.knn_point_major(xp.detach().view(B, N, 3), k).astype(np.int64))

    rc = [t.clone() for t in rms]
    x = xp.view(B, N, 3).transpose(2, 1)
    f = ref_cpu.graph_feature(x, idx)
    y = torch.einsum("oc,bcnk->bonk", W1, f)
    y = F.leaky_relu(F.batch_norm(y, rc[0], rc[1], g1, b1, training, 0.1, 1e-5), 0.2)
    z = torch.einsum("oc,bcnk->bonk", W2, y)
    z = F.leaky_relu(F.batch_norm(z, rc[2], rc[3], g2, b2, training, 0.1, 1e-5), 0.2)
    oc = z.max(dim=-1)[0].transpose(2, 1).reshape(P, 128)
    oc.backward(dOut)

    leaves = [xp, W1, g1, b1, W2, g2, b2]
    gl = [t.detach().to(dev).requires_grad_(True) for t in leaves]
    rg = [t.to(dev) for t in rms]
    if k <= 40:
        graph = Fh.knn_graph(gl[0], B, N, k)
        assert np.array_equal(graph.idx.view(B, N, k).cpu().numpy(), idx.numpy())
    else:                       # the kNN kernels stop at k = 40 for 3 channels; the fused stage itself takes any k <= 128
        graph = Fh.graph_from_indices(idx.to(dev), B, N, k)
    og = Fh.tnet_edge(gl[0], graph, gl[1], gl[2], gl[3], rg[0], rg[1], gl[4], gl[5], gl[6], rg[2], rg[3], training)
    og.backward(dOut.to(dev))
    np.testing.assert_allclose(og.detach().cpu().numpy(), oc.detach().numpy(), rtol=3e-4, atol=3e-4)
    for got, want, name in zip([t.grad for t in gl], [t.grad for t in leaves], ["dx", "dW1", "dg1", "db1", "dW2", "dg2", "db2"]):
        scale = want.abs().max().item() + 1e-6
        err = (got.cpu() - want).abs().max().item()
        assert err / scale < 3e-3, (name, err, scale)
    if training:
        for a, b in zip(rg, rc):
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-4, atol=1e-5)
    # the input cloud without a gradient (how DGCNN calls it): dW1 comes from the sequential moment pass (tnet_bwd_tmom_kernel) instead of
    # the fold onto the points + GEMM -- same gradients
    g2l = [t.detach().to(dev).requires_grad_(i > 0) for i, t in enumerate(leaves)]
    rg2 = [t.to(dev) for t in rms]
    og2 = Fh.tnet_edge(g2l[0], graph, g2l[1], g2l[2], g2l[3], rg2[0], rg2[1], g2l[4], g2l[5], g2l[6], rg2[2], rg2[3], training)
    og2.backward(dOut.to(dev))
    assert torch.equal(og2, og) and g2l[0].grad is None
    for a, b, want, name in zip([t.grad for t in g2l[1:]], [t.grad for t in gl[1:]], [t.grad for t in leaves[1:]], ["dW1", "dg1", "db1", "dW2", "dg2", "db2"]):
        scale = want.abs().max().item() + 1e-6
        assert (a.cpu() - want).abs().max().item() / scale < 3e-3, name
        assert (a - b).abs().max().item() / scale < (2e-4 if name == "dW1" else 1e-7), (name, (a - b).abs().max().item() / scale)


@pytest.mark.parametrize("mode", ["f16x3", "bf16"])
@pytest.mark.parametrize("M", [16, 32, 7])
def test_per_cloud_layers_sharing_an_input_gradient(dev, M, mode):
    """The x5 halves of the PointSegDA heads' first layers (PointSegDA/Models.py:226-241: every head reads the same [B, 1024] global
    feature): per-cloud Linear layers (rows = clouds <= 32) whose input gradients are SUMMED in one buffer (functional.fan_out /
    grad_accum: beta = 1 in the dgrad).  Round 6: that dgrad runs on skinny_bwd_pair_kernel (dx_accumulate) instead of the bounds-checked
    128 x 128 tile kernel.  Against torch: outputs, the summed input gradient, every weight gradient."""
    Fh = _fh()
    x = _rand((M, 1024), 31)
    Ws = [_rand((n, 1024), 32 + i, 0.05) for i, n in enumerate((256, 256, 512))]
    dYs = [_rand((M, W.shape[0]), 40 + i) for i, W in enumerate(Ws)]
    xc = x.clone().requires_grad_(True)
    Wc = [W.clone().requires_grad_(True) for W in Ws]
    torch.autograd.backward([xc @ W.t() for W in Wc], dYs)
    xg = x.to(dev).requires_grad_(True)
    Wg = [W.to(dev).requires_grad_(True) for W in Ws]
    with Fh.gemm_precision(mode):
        aliases, acc = Fh.fan_out(xg, len(Wg))
        outs = [Fh.pointmlp(a, W, grad_accum=acc) for a, W in zip(aliases, Wg)]
        torch.autograd.backward(outs, [d.to(dev) for d in dYs])
    for o, W in zip(outs, Wc):
        np.testing.assert_allclose(o.detach().cpu().numpy(), (x @ W.detach().t()).numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-4, atol=1e-4)
    for a, b in zip(Wg, Wc):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,N,k", [(4, 512, 20), (2, 1024, 40)])
def test_tnet_edge_bf16_operands_vs_fp32_products(dev, B, N, k):
    """`precision` "bf16" (BASELINE.json configs[4]): the per-edge stage multiplies operands ROUNDED to bf16 -- one MFMA per k16 step of the
    64 -> 128 contraction and of the backward's Gram-form products (tnet.hip, ONEP instantiations) where the fp32-accurate modes spend six.
    Forward: against the fp32-accurate mode of the same call.  Backward: the SAME forward (bf16 mode: same saved pre-activations and
    arg-max slots) taken back once with the single products and once with the six -- a rounding of both operands apart (2^-9 per value),
    without the max-over-k selections a rounded forward moves (those are what tests/test_gpu_model.py's yardstick prices at model level).
    Not less than a rounding leaves either: the operands really were rounded."""
    from mlsp_amd import _lib
    Fh = _fh()
    P = B * N
    g = torch.Generator().manual_seed(21)
    xp = torch.randn(P, 3, generator=g).to(dev)
    W1, W2 = (torch.randn(64, 6, generator=g) * 0.5).to(dev), (torch.randn(128, 64, generator=g) * 0.2).to(dev)
    g1, b1 = (torch.randn(64, generator=g) * 0.3 + 1).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
    g2, b2 = (torch.randn(128, generator=g) * 0.3 + 1).to(dev), (torch.randn(128, generator=g) * 0.3).to(dev)
    dOut = torch.randn(P, 128, generator=g).to(dev)
    graph = Fh.knn_graph(xp, B, N, k)

    def run(fwd_mode, bwd_mode):
        leaves = [t.clone().requires_grad_(True) for t in (W1, g1, b1, W2, g2, b2)]
        rm1, rv1, rm2, rv2 = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(128, device=dev), torch.ones(128, device=dev)
        with Fh.gemm_precision(fwd_mode):
            out = Fh.tnet_edge(xp, graph, leaves[0], leaves[1], leaves[2], rm1, rv1, leaves[3], leaves[4], leaves[5], rm2, rv2, True)
        assert out.grad_fn.prec == _lib.GEMM_PRECISION_MODES[fwd_mode]
        out.grad_fn.prec = _lib.GEMM_PRECISION_MODES[bwd_mode]        # (the Function hands its forward's mode to its backward: overridden here)
        out.backward(dOut)
        return [out.detach()] + [t.grad for t in leaves]

    ref_f, one, six = run("bf16x6", "bf16x6"), run("bf16", "bf16"), run("bf16", "bf16x6")
    rel_out = ((one[0].double() - ref_f[0].double()).norm() / ref_f[0].double().norm()).item()
    assert 1e-5 < rel_out < 1e-2, rel_out
    assert torch.equal(one[0], six[0])
    for i, name in enumerate(("dW1", "dg1", "db1", "dW2", "dg2", "db2"), 1):
        a, b = one[i].double(), six[i].double()
        assert torch.isfinite(a).all(), name
        rel = ((a - b).norm() / b.norm()).item()
        assert rel < 1e-2, (name, rel)
        if name == "dW2":
            assert rel > 1e-5, (name, rel)


@pytest.mark.parametrize("B,N,k", [(8, 512, 20), (4, 512, 40)])
def test_tnet_backward_gram_form_vs_round1_kernel(dev, B, N, k):
    """Three independent backward kernels of the fused stage: the dense split-product Gram form (the default), the f32 Gram form with the
    register-indexed sparse part (MLSP_TNET_BWD_F32=1) and the round-1 kernel (three MFMA products on the dZ tile, MLSP_TNET_BWD_OLD=1);
    the switches are read once per process, hence the subprocesses."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("cmp_tnet_bwd", os.path.join(os.path.dirname(__file__), "..", "tools", "cmp_tnet_bwd.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, err in mod.compare(B, N, k).items():
        assert err < 1e-4, (name, err)


@pytest.mark.parametrize("fused_stats", [False, True])
@pytest.mark.parametrize("mode", ["fp32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("M,C0,C1,C2,training,clouds", [(4096, 128, 256, 128, True, 0), (8192, 512, 256, 256, True, 0), (1000, 96, 80, 64, True, 0),
                                                        (4096, 128, 256, 128, False, 0), (16384, 256, 1024, 512, True, 0),
                                                        (2048, 512, 512, 256, True, 0), (8192, 128, 256, 256, True, 8), (4096, 64, 512, 256, True, 16)])
def test_pointmlp_deferred_activation_chain(dev, M, C0, C1, C2, training, clouds, mode, fused_stats, monkeypatch):
    """pointmlp(..., chain=True) under fp32 storage hands its PRE-BN output to the next layer, which applies BN + ReLU + dropout in
    its GEMM operand loads (forward: A rows, wgrad: the k-major B operand) -- gemm_split_kernel<.., XF, XD> in mode "bf16x6", the f32
    transform kernels in mode "fp32" and on short K loops, the streaming kernels of thin.hip for the 3-channel output layer --
    BIT-IDENTICAL to the materialised path: the staged values are computed by the same expressions, the products by the same kernels.
    The third case is outside the interior-tile path (one streaming pass into the workspace instead).  clouds > 0: the first layer takes a
    per-cloud bias row (the heads' first layer: Models.py:192); with fused statistics its gradient comes from the row-panel sums."""
    Fh = _fh()
    import itertools as it

    torch.manual_seed(77)
    monkeypatch.setattr(Fh, "_FUSE_BWD_STATS", fused_stats)

    def run(defer):
        monkeypatch.setattr(Fh, "_DEFER_CHAINS", defer)
        monkeypatch.setattr(Fh, "_seed_counter", it.count(1234), raising=False)
        ts = [(_rand((M, C0), 1)), _rand((C1, C0), 2, 0.2), _rand((C2, C1), 3, 0.2), _rand((3, C2), 4, 0.2)]
        gb = [(_rand((C1,), 5) + 0.3), _rand((C1,), 6), (_rand((C2,), 7) + 0.3), _rand((C2,), 8)]
        leaves = [t.to(dev).requires_grad_(True) for t in ts + gb]
        X, W1, W2, W3, g1, b1, g2, b2 = leaves
        cb = _rand((clouds, C1), 10).to(dev).requires_grad_(True) if clouds else None
        rs = [torch.zeros(C1, device=dev), torch.ones(C1, device=dev), torch.zeros(C2, device=dev), torch.ones(C2, device=dev)]
        h = Fh.pointmlp(X, W1, gbias=cb, rows_per_group=M // clouds if clouds else 0, gamma=g1, beta=b1, run_mean=rs[0], run_var=rs[1],
                        training=training, act=Fh.ACT_RELU, p_drop=0.5, chain=True)
        assert isinstance(h, Fh.DeferredAct) == bool(defer)
        h = Fh.pointmlp(h, W2, gamma=g2, beta=b2, run_mean=rs[2], run_var=rs[3], training=training, act=Fh.ACT_LRELU, slope=0.2, p_drop=0.3,
                        chain=True)
        out = Fh.pointmlp(h, W3, training=training)
        out.backward(_rand((M, 3), 9).to(dev))
        return [out.detach().cpu()] + [t.grad.cpu() for t in leaves] + [r.cpu() for r in rs] + ([cb.grad.cpu()] if clouds else [])

    with Fh.gemm_precision(mode):
        a, b = run(True), run(False)
    names = ["out", "dX", "dW1", "dW2", "dW3", "dg1", "db1", "dg2", "db2", "rm1", "rv1", "rm2", "rv2", "dcloudbias"]
    for n, x, y in zip(names, a, b):
        if mode == "f16x3":
            # two-piece f16 products: the deferred path scales a transformed operand by its ANALYTIC bound (batch statistics), the
            # materialised path by the measured magnitude -- another power of two: the same pieces except for elements that reach the f16
            # subnormals under one scale and not the other (< 2^-40 of the operand's largest magnitude).  Rounding-level agreement.
            err = (x - y).double().norm().item()
            assert err <= 2e-5 * max(y.double().norm().item(), 1e-3 * a[names.index("dg" + n[-1])].double().norm().item() if n.startswith("db") else 1e-30), (n, err)
        elif not fused_stats or n in ("out", "rm1", "rv1", "rm2", "rv2"):
            assert torch.equal(x, y), (n, (x - y).abs().max().item())
        else:
            # fused statistics (the consumers' dgrads leave the producer's BatchNorm-backward sums: gemm_out_bs): the masked gradient is
            # bit-identical, its column sums are accumulated in another order (fp32 inside a wave, then fp64): ~1e-7 relative on the
            # coefficients; db (a sum of ~cancelling terms) is compared against the gradient scale
            err = (x - y).double().norm().item()
            assert err <= 2e-5 * max(y.double().norm().item(), 1e-3 * a[names.index("dg" + n[-1])].double().norm().item() if n.startswith("db") else 0.0), (n, err)


@pytest.mark.parametrize("fused_stats", [False, True])
@pytest.mark.parametrize("mode", ["bf16x6", "fp32", "f16x3"])
@pytest.mark.parametrize("M,training", [(2048, True), (16384, True), (4096, False), (1100, True)])
def test_merged_layers_deferred_activation_chain(dev, M, training, mode, fused_stats, monkeypatch):
    """The head stack of PointDA/Models.py:192-197, 226-231, 272-285 as the model runs it: one wide first layer (pointmlp), two merged
    depths (multimlp: block-diagonal launch for the two identical region-head segments + one single segment, per-channel activation and
    dropout), the three final Linear layers on column slices (thin kernels).  With deferred activations NO layer writes its activated
    output: every consumer -- single, grouped and k-major weight-gradient GEMM launches, the thin kernels -- transforms its operand,
    slices of a merged layer keep their place in the producer's dropout stream.  Bit-identical to the materialised chain (outputs,
    every gradient, running statistics).  M = 1100: shapes outside the interior-tile kernels take the streaming fallback."""
    Fh = _fh()
    import itertools as it
    C0 = 512
    torch.manual_seed(1234)          # (the dropout streams are keyed by torch.initial_seed(), which is random per process by default)
    monkeypatch.setattr(Fh, "_FUSE_BWD_STATS", fused_stats)

    def run(defer):
        monkeypatch.setattr(Fh, "_DEFER_CHAINS", defer)
        monkeypatch.setattr(Fh, "_seed_counter", it.count(4321), raising=False)
        g = torch.Generator().manual_seed(5)
        def r(*shape, s=1.0):
            return ((torch.rand(shape, generator=g) * 2 - 1) * s).to(dev).requires_grad_(True)
        X = r(M, C0)
        W1, g1, b1 = r(1024, C0, s=0.1), r(1024), r(1024)
        d0 = [(r(256, 256, s=0.1), None), (r(256, 256, s=0.1), None), (r(256, 512, s=0.1), r(256))]      # (W, bias) per segment
        d1 = [(r(128, 256, s=0.1), None), (r(128, 256, s=0.1), None), (r(256, 256, s=0.1), r(256))]
        gb0, gb1 = (r(768), r(768)), (r(512), r(512))
        fin = [(r(3, 128, s=0.2), None), (r(3, 128, s=0.2), None), (r(16, 256, s=0.2), r(16))]
        stats = [torch.zeros(1024, device=dev), torch.ones(1024, device=dev), torch.zeros(768, device=dev), torch.ones(768, device=dev),
                 torch.zeros(512, device=dev), torch.ones(512, device=dev)]
        h = Fh.pointmlp(X, W1, gamma=g1, beta=b1, run_mean=stats[0], run_var=stats[1], training=training, act=Fh.ACT_RELU, p_drop=0.5, chain=True)
        assert isinstance(h, Fh.DeferredAct) == bool(defer)
        spec0 = ((256, 0.0, True), (256, 0.0, True), (256, 0.2, True))
        spec1 = ((128, 0.0, False), (128, 0.0, False), (256, 0.2, True))
        h = Fh.multimlp(h, [(0, d0[0][0], d0[0][1]), (256, d0[1][0], d0[1][1]), (512, d0[2][0], d0[2][1])], gb0[0], gb0[1], stats[2], stats[3],
                        Fh.channel_params(dev, spec0), training=training, p_drop=0.5, chain=True, spec=spec0)
        assert isinstance(h, Fh.DeferredAct) == bool(defer)
        h = Fh.multimlp(h, [(0, d1[0][0], d1[0][1]), (256, d1[1][0], d1[1][1]), (512, d1[2][0], d1[2][1])], gb1[0], gb1[1], stats[4], stats[5],
                        Fh.channel_params(dev, spec1), training=training, p_drop=0.5, chain=True, spec=spec1)
        slices, cols = Fh.split_columns_shared(h, [128, 128, 256])
        outs, col = [], 0
        for sl, (W, b) in zip(slices, fin):
            outs.append(Fh.pointmlp(sl, W, bias=b, training=training, grad_cols=(cols, col)))
            col += sl.shape[1]
        loss = sum((o * _rand(tuple(o.shape), 40 + i).to(dev)).sum() for i, o in enumerate(outs))
        loss.backward()
        leaves = [X, W1, g1, b1] + [t for pair in d0 + d1 + fin for t in pair if t is not None] + list(gb0) + list(gb1)
        return [o.detach().cpu() for o in outs] + [t.grad.cpu() for t in leaves] + [s.cpu() for s in stats]

    with Fh.gemm_precision(mode):
        a, b = run(True), run(False)
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        if mode == "bf16x6" and (not fused_stats or i < 3 or i >= len(a) - 6):
            assert torch.equal(x, y), (i, tuple(x.shape), (x - y).abs().max().item())
        elif mode == "bf16x6":
            # fused BatchNorm-backward statistics: same masked gradients, column sums in another order (see the test above)
            rel = ((x - y).double().norm() / (y.double().norm() + 1e-30)).item()
            assert rel < 1e-4, (i, tuple(x.shape), rel)
        elif mode == "f16x3":
            # (another power-of-two operand scale on the deferred path: see test_pointmlp_deferred_activation_chain; a last-bit difference
            # in a pre-activation can flip a ReLU unit, as in mode "fp32" below)
            rel = ((x - y).double().norm() / (y.double().norm() + 1e-30)).item()
            assert rel < (1e-5 if i < 3 or i >= len(a) - 6 else 5e-3), (i, tuple(x.shape), rel)
        else:
            # mode "fp32": the f32 transform kernels take no block-diagonal launch, so the deferred path runs the two region-head segments
            # one by one -- with another row-panel height, i.e. another grouping of the BatchNorm partial sums: last-bit differences
            # ... and a last-bit difference in a pre-activation flips a ReLU for a handful of the 10^7 elements, which moves a whole
            # gradient row: outputs / statistics at rounding level, gradients at the level of that re-routing
            rel = ((x - y).double().norm() / (y.double().norm() + 1e-30)).item()
            assert rel < (1e-5 if i < 3 or i >= len(a) - 6 else 5e-3), (i, tuple(x.shape), rel)


@pytest.mark.parametrize("subset", [(0,), (2,), (0, 2), (0, 1)])
@pytest.mark.parametrize("mode", ["bf16x6", "fp32", "f16x3"])
def test_deferred_chain_loss_on_a_subset_of_the_consumers(dev, subset, mode, monkeypatch):
    """The reference trainer runs all three heads (activate_density_normal_ondef=True) and gates the Normal / density losses on
    args.Normal_ondef / args.Density_ondef (PointDA/trainer.py:551-565): the backward then reaches only SOME of the final Linear layers.
    The consumers that do run store their input gradient masked by the producer's activation derivative and dropout (fused
    BatchNorm-backward statistics, BwdStats); the producer must neither apply the mask a second time (dropout columns would carry
    1 / (1 - p)^2, LeakyReLU slopes squared) nor use the incomplete sums: it takes the masked-without-sums path (pre_parts = -1,
    include/mlsp_hip.h).  Against the materialised, unfused chain (the bit-identical path of the two tests above)."""
    Fh = _fh()
    import itertools as it
    M, C0, training = 4096, 512, True
    torch.manual_seed(4321)

    def run(defer, fused):
        monkeypatch.setattr(Fh, "_FUSE_BWD_STATS", fused)
        monkeypatch.setattr(Fh, "_DEFER_CHAINS", defer)
        monkeypatch.setattr(Fh, "_seed_counter", it.count(99), raising=False)
        g = torch.Generator().manual_seed(6)
        def r(*shape, s=1.0):
            return ((torch.rand(shape, generator=g) * 2 - 1) * s).to(dev).requires_grad_(True)
        X = r(M, C0)
        W1, g1, b1 = r(1024, C0, s=0.1), r(1024), r(1024)
        d0 = [(r(256, 256, s=0.1), None), (r(256, 256, s=0.1), None), (r(256, 512, s=0.1), r(256))]
        d1 = [(r(128, 256, s=0.1), None), (r(128, 256, s=0.1), None), (r(256, 256, s=0.1), r(256))]
        gb0, gb1 = (r(768), r(768)), (r(512), r(512))
        fin = [(r(3, 128, s=0.2), None), (r(3, 128, s=0.2), None), (r(16, 256, s=0.2), r(16))]
        stats = [torch.zeros(1024, device=dev), torch.ones(1024, device=dev), torch.zeros(768, device=dev), torch.ones(768, device=dev),
                 torch.zeros(512, device=dev), torch.ones(512, device=dev)]
        h = Fh.pointmlp(X, W1, gamma=g1, beta=b1, run_mean=stats[0], run_var=stats[1], training=training, act=Fh.ACT_RELU, p_drop=0.5, chain=True)
        spec0 = ((256, 0.0, True), (256, 0.0, True), (256, 0.2, True))
        spec1 = ((128, 0.2, True), (128, 0.0, True), (256, 0.2, True))      # (dropout + LeakyReLU in front of the final layers: the double mask would show)
        h = Fh.multimlp(h, [(0, d0[0][0], d0[0][1]), (256, d0[1][0], d0[1][1]), (512, d0[2][0], d0[2][1])], gb0[0], gb0[1], stats[2], stats[3],
                        Fh.channel_params(dev, spec0), training=training, p_drop=0.5, chain=True, spec=spec0)
        h = Fh.multimlp(h, [(0, d1[0][0], d1[0][1]), (256, d1[1][0], d1[1][1]), (512, d1[2][0], d1[2][1])], gb1[0], gb1[1], stats[4], stats[5],
                        Fh.channel_params(dev, spec1), training=training, p_drop=0.5, chain=True, spec=spec1)
        assert isinstance(h, Fh.DeferredAct) == bool(defer)
        slices, cols = Fh.split_columns_shared(h, [128, 128, 256])
        outs, col = [], 0
        for sl, (W, b) in zip(slices, fin):
            outs.append(Fh.pointmlp(sl, W, bias=b, training=training, grad_cols=(cols, col)))
            col += sl.shape[1]
        loss = sum((outs[i] * _rand(tuple(outs[i].shape), 40 + i).to(dev)).sum() for i in subset)
        loss.backward()
        leaves = [X, W1, g1, b1] + [t for pair in d0 + d1 for t in pair if t is not None] + list(gb0) + list(gb1)
        return [t.grad.cpu() for t in leaves]

    with Fh.gemm_precision(mode):
        a, b = run(True, True), run(False, False)
    for i, (x, y) in enumerate(zip(a, b)):
        rel = ((x - y).double().norm() / (y.double().norm() + 1e-30)).item()
        assert rel < (1e-4 if mode == "bf16x6" else 5e-3), (i, tuple(x.shape), rel)


# ----------------------------------------------------------------------------- conv + BN + act + max over N (Gram backward)
@pytest.mark.parametrize("B,N,Cin,Cout,training", [(4, 100, 64, 96, True), (3, 128, 128, 256, True), (2, 64, 512, 1024, True),
                                                   (3, 50, 40, 70, False)])
def test_pointmlp_colmax_fwd_bwd(dev, B, N, Cin, Cout, training):
    Fh = _fh()
    P = B * N
    X = _rand((P, Cin), 1).requires_grad_(True)
    W = _rand((Cout, Cin), 2, 0.2).requires_grad_(True)
    gamma = (_rand((Cout,), 3) + 0.3).requires_grad_(True)
    beta = _rand((Cout,), 4).requires_grad_(True)
    rm, rv = _rand((Cout,), 5) * 0.1, _rand((Cout,), 6).abs() + 0.5
    dOut = _rand((B, Cout), 7)
    rm_c, rv_c = rm.clone(), rv.clone()
    Y = F.batch_norm(X @ W.t(), rm_c, rv_c, gamma, beta, training, 0.1, 1e-5)
    oc = F.leaky_relu(Y, 0.2).view(B, N, Cout).max(dim=1)[0]
    oc.backward(dOut)
    leaves = [X, W, gamma, beta]
    gl = [t.detach().to(dev).requires_grad_(True) for t in leaves]
    rm_g, rv_g = rm.to(dev), rv.to(dev)
    og = Fh.pointmlp_colmax(gl[0], gl[1], gl[2], gl[3], rm_g, rv_g, B, N, training=training)
    og.backward(dOut.to(dev))
    np.testing.assert_allclose(og.detach().cpu().numpy(), oc.detach().numpy(), rtol=2e-4, atol=2e-4)
    for got, want, name in zip([t.grad for t in gl], [t.grad for t in leaves], ["dX", "dW", "dgamma", "dbeta"]):
        scale = want.abs().max().item() + 1e-6
        err = (got.cpu() - want).abs().max().item()
        assert err / scale < 3e-3, (name, err, scale)
    if training:
        np.testing.assert_allclose(rm_g.cpu().numpy(), rm_c.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rv_g.cpu().numpy(), rv_c.numpy(), rtol=1e-4, atol=1e-5)


def test_knn_two_pass_overflow_fallback(dev):
    """v4 threshold select: massive ties overflow the survivor buffers and must fall back to the exact
    sequential path (all-identical points; few distinct points; a cloud that mixes both with random points)."""
    Fh = _fh()
    B, N, k = 3, 512, 20
    x = torch.zeros(B, N, 3)
    x[1] = (torch.arange(N) % 4).float().view(N, 1).expand(N, 3) * 0.25      # 4 distinct points, 128 copies each
    x[2] = _rand((N, 3), 9)
    x[2, 100:400] = x[2, 100:101]                                            # 300 duplicates inside a random cloud
    want = knn_canon.knn_point_major(x, k)
    got = Fh.knn_graph(x.view(B * N, 3).to(dev), B, N, k).idx.view(B, N, k).cpu().numpy()
    assert np.array_equal(got, want), int((got != want).any(-1).sum())
    xf = _rand((2 * 1024, 64), 3)
    xf[:1024, :] = xf[:1, :]                                                 # feature-space stage, one cloud degenerate
    want = knn_canon.knn_point_major(xf.view(2, 1024, 64), k)
    got = Fh.knn_graph(xf.to(dev), 2, 1024, k).idx.view(2, 1024, k).cpu().numpy()
    assert np.array_equal(got, want)


# ----------------------------------------------------------------------------- API parity of the op layer's optional arguments
@pytest.mark.parametrize("training", [True, False])
def test_conv2d_edge_with_bias_and_fc_without_bn(dev, training):
    """model_utils.conv_2d(bias=True) on the fused EdgeConv path and fc_layer(bn=False) (PointDA/model_utils.py:45-87 accept both):
    compared with the same modules' materialised reference-shaped forward (graph feature -> conv+BN+act -> max over k) / plain torch."""
    from mlsp_amd import model_utils as mu
    Fh = _fh()
    torch.manual_seed(3)
    B, N, C, k = 2, 128, 8, 20
    conv = mu.conv_2d(2 * C, 32, 1, activation='leakyrelu', bias=True).to(dev)
    with torch.no_grad():
        conv.conv[0].bias.normal_(0, 0.5)
        conv.conv[1].weight.uniform_(0.5, 1.5)
        conv.conv[1].bias.normal_(0, 0.2)
        conv.conv[1].running_mean.normal_(0, 0.3)
        conv.conv[1].running_var.uniform_(0.5, 1.5)
    conv.train(training)
    ref = __import__("copy").deepcopy(conv)
    x = _rand((B, C, N), 9).to(dev)
    xp = x.transpose(2, 1).contiguous().view(B * N, C).requires_grad_(True)
    g = Fh.knn_graph(xp, B, N, k)
    got = conv.edge(xp, g)
    got.square().sum().backward()
    xr = x.clone().requires_grad_(True)
    feat = mu.get_graph_feature(xr, None, k=k, idx=g.idx.view(B, N, k).long())            # [B,2C,N,k]
    want = ref(feat).max(dim=-1)[0].permute(0, 2, 1).reshape(B * N, 32)
    want.square().sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(xp.grad.cpu().numpy(), xr.grad.transpose(2, 1).reshape(B * N, C).cpu().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(conv.conv[0].weight.grad.cpu().numpy(), ref.conv[0].weight.grad.cpu().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(conv.conv[1].running_mean.cpu().numpy(), ref.conv[1].running_mean.cpu().numpy(), rtol=1e-4, atol=1e-5)
    if training:
        assert conv.conv[0].bias.grad is not None and conv.conv[0].bias.grad.abs().max().item() == 0.0
    fc = mu.fc_layer(64, 48, bn=False, activation='leakyrelu').to(dev).train(training)
    xin = _rand((40, 64), 4).to(dev).requires_grad_(True)
    out = fc(xin)
    out.sum().backward()
    w, b = fc.fc[0].weight.detach(), fc.fc[0].bias.detach()
    np.testing.assert_allclose(out.detach().cpu().numpy(), F.leaky_relu(xin.detach() @ w.t() + b, 0.2).cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert fc.fc[0].weight.grad is not None and xin.grad is not None


def test_apply_transform_vs_bmm(dev):
    """functional.apply_transform (the 3x3 input transform of Models.py:113 on point-major rows) against torch.bmm, forward and backward."""
    Fh = _fh()
    B, N = 5, 333
    x = _rand((B * N, 3), 31).to(dev).requires_grad_(True)
    T = (_rand((B, 3, 3), 32) + torch.eye(3)).to(dev).requires_grad_(True)
    w = _rand((B * N, 3), 33).to(dev)
    out = Fh.apply_transform(x, T)
    (out * w).sum().backward()
    gx, gT = x.grad.clone(), T.grad.clone()
    x.grad = T.grad = None
    ref = torch.bmm(x.view(B, N, 3), T.transpose(1, 2)).view(B * N, 3)
    (ref * w).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(gx.cpu().numpy(), x.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gT.cpu().numpy(), T.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)


# ----------------------------------------------------------------------------- Linear + BN + act + max over every k rows, fused
@pytest.mark.parametrize("G,k,Cin,Cout,act,training,use_bias", [(64, 32, 64, 128, 1, True, True), (48, 20, 96, 256, 2, True, False),
                                                                (16, 128, 131, 64, 1, False, True), (200, 7, 32, 1024, 0, True, True)])
def test_pointmlp_segmax_vs_torch(dev, G, k, Cin, Cout, act, training, use_bias):
    """mlsp_pointmlp_segmax_*_f32 (the activated [G*k, Cout] tensor is never written; BN-backward sums from the selected entries only)
    against Linear -> BatchNorm1d -> activation -> max over every k consecutive rows in torch on the CPU; negative BN scales included
    (min-select path), eval mode, first-occurrence ties irrelevant for random data."""
    Fh = _fh()
    M = G * k
    X = _rand((M, Cin), 1).requires_grad_(True)
    W = _rand((Cout, Cin), 2, 0.2).requires_grad_(True)
    b = _rand((Cout,), 3).requires_grad_(True) if use_bias else None
    g = (_rand((Cout,), 4) * 0.8 + 0.1).requires_grad_(True)            # both signs
    be = _rand((Cout,), 5).requires_grad_(True)
    rm, rv = _rand((Cout,), 6) * 0.1, _rand((Cout,), 7).abs() + 0.5
    dOut = _rand((G, Cout), 8)
    rmc, rvc = rm.clone(), rv.clone()
    Y = X @ W.t() + (b if use_bias else 0.0)
    Z = F.batch_norm(Y, rmc, rvc, g, be, training, 0.1, 1e-5)
    Z = F.relu(Z) if act == 1 else F.leaky_relu(Z, 0.2) if act == 2 else Z
    oc = Z.view(G, k, Cout).max(dim=1)[0]
    oc.backward(dOut)
    leaves = [t for t in (X, W, b, g, be) if t is not None]
    gl = [t.detach().to(dev).requires_grad_(True) for t in leaves]
    it = iter(gl)
    Xg, Wg = next(it), next(it)
    bg = next(it) if use_bias else None
    gg, beg = next(it), next(it)
    rmg, rvg = rm.to(dev), rv.to(dev)
    og = Fh.pointmlp_segmax(Xg, Wg, k, bias=bg, gamma=gg, beta=beg, run_mean=rmg, run_var=rvg, training=training, act=act, slope=0.2)
    og.backward(dOut.to(dev))
    np.testing.assert_allclose(og.detach().cpu().numpy(), oc.detach().numpy(), rtol=2e-4, atol=2e-4)
    for got, want, name in zip([t.grad for t in gl], [t.grad for t in leaves], [n for n, t in zip(["dX", "dW", "db", "dg", "dbe"], (X, W, b, g, be)) if t is not None]):
        if name == "db" and training:
            assert got.abs().max().item() == 0.0                       # a bias in front of batch statistics: analytically zero
            continue
        scale = want.abs().max().item() + 1e-6
        assert (got.cpu() - want).abs().max().item() / scale < 2e-3, (name, (got.cpu() - want).abs().max().item() / scale)
    if training:
        np.testing.assert_allclose(rmg.cpu().numpy(), rmc.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rvg.cpu().numpy(), rvc.numpy(), rtol=1e-4, atol=1e-5)


def test_gemm_fold64_weight_gradient(dev):
    """A 64 x 64 weight gradient over many rows (SA level 1, conv2: dW = dY^T X over 524,288 edges) runs as ONE 128 x 128 interior-tile launch
    over row PAIRS plus a 64 x 64 sum of its diagonal blocks (gemm.hip gemm_fold64): same result as the float64 product to fp32 accuracy,
    in both product modes, and the non-foldable neighbours (pitch != 64, short K) keep their old path."""
    Fh = _fh()
    g = torch.Generator().manual_seed(5)
    for K in (8192, 65536, 524288):
        A, B = torch.randn(K, 64, generator=g).to(dev), torch.randn(K, 64, generator=g).to(dev)
        ref = A.double().t() @ B.double()
        for mode in ("fp32", "bf16x6"):
            with Fh.gemm_precision(mode):
                got = Fh.gemm(A, B, True, False).double()
            rel = ((got - ref).norm() / ref.norm()).item()
            assert rel < 2e-6, (K, mode, rel)
    A, B = torch.randn(8192, 96, generator=g).to(dev), torch.randn(8192, 64, generator=g).to(dev)
    got = Fh.gemm(A[:, :64], B, True, False).double()                      # pitch 96: not foldable
    assert ((got - A[:, :64].double().t() @ B.double()).norm() / got.norm()).item() < 2e-6


def test_gemm_split_layouts_agree_and_are_linear(dev):
    """Size-independent properties of the split kernel at the configs[1] layer size: the four operand layouts (row-major images / k-major
    images read with ds_read_b64_tr_b16, in either operand) compute the same product to fp32 accuracy, and the kernel is linear in an
    operand (C(A1 + A2, B) = C(A1, B) + C(A2, B) to fp32 accuracy; exactly so for a power-of-two scale)."""
    Fh = _fh()
    M, N, K = 32768, 512, 1024
    A, B = _rand((M, K), 21).to(dev), _rand((N, K), 22).to(dev)
    At, Bt = A.t().contiguous(), B.t().contiguous()
    with Fh.gemm_precision("bf16x6"):
        c_nt = Fh.gemm(A, B, False, True)
        c_nn = Fh.gemm(A, Bt, False, False)
        c_tt = Fh.gemm(At, B, True, True)
        c_tn = Fh.gemm(At, Bt, True, False)
        scale = c_nt.double().norm()
        for other in (c_nn, c_tt, c_tn):
            assert ((other.double() - c_nt.double()).norm() / scale).item() < 5e-7
        assert torch.equal(Fh.gemm(A * 4.0, B, False, True), c_nt * 4.0)            # the split of 4x is 4 x the split of x
        A2 = _rand((M, K), 23).to(dev)
        lin = Fh.gemm(A + A2, B, False, True).double() - (c_nt.double() + Fh.gemm(A2, B, False, True).double())
        assert (lin.norm() / scale).item() < 1e-6


def test_tnet_forward_accuracy_both_product_kernels(dev):
    """The fused T-Net per-edge stage against a float64 evaluation of the reference's op sequence at (B=8, N=1024, k=20) and (4, 2048, 40):
    relative L2 <= 4e-7 for the f32-MFMA kernel (MLSP_TNET_FWD_F32=1) and for the split-products kernel (MLSP_TNET_FWD_SPLIT=1; the default from
    1024 tiles up, i.e. at both of these shapes).  The switches are read once per process, hence the subprocesses."""
    import subprocess, sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    for env_extra in ({"MLSP_TNET_FWD_F32": "1"}, {"MLSP_TNET_FWD_SPLIT": "1"}):
        env = {k: v for k, v in os.environ.items() if k not in ("MLSP_TNET_FWD_F32", "MLSP_TNET_FWD_SPLIT")}
        env.update(env_extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "tnet_acc.py"), "--check"], env=env, capture_output=True, text=True, timeout=300)
        print(env_extra, r.stdout.strip().splitlines()[-2:])
        assert r.returncode == 0, (env_extra, r.stdout[-500:], r.stderr[-500:])


@pytest.mark.parametrize("Cm,Ci,Co", [(64, 6, 64), (64, 128, 64), (7, 3, 5), (130, 257, 33)])
def test_compose_linear_fwd_bwd(dev, Cm, Ci, Co):
    """PointSegDA/Models.py:176-182: conv_b(conv_a(f)) as one linear map; values and all four parameter gradients against float64."""
    Fh = _fh()
    Wa, ba, Wb, bb = (_rand(sh, 900 + i).to(dev).requires_grad_() for i, sh in enumerate(((Cm, Ci), (Cm,), (Co, Cm), (Co,))))
    W, b = Fh.compose_linear(Wa, ba, Wb, bb)
    gW, gb = _rand((Co, Ci), 910).to(dev), _rand((Co,), 911).to(dev)
    torch.autograd.backward([W, b], [gW, gb])
    ref = [t.detach().double().cpu().requires_grad_() for t in (Wa, ba, Wb, bb)]
    Wr, br = ref[2] @ ref[0], ref[2] @ ref[1] + ref[3]
    torch.autograd.backward([Wr, br], [gW.double().cpu(), gb.double().cpu()])
    for got, want in ((W, Wr), (b, br), (Wa.grad, ref[0].grad), (ba.grad, ref[1].grad), (Wb.grad, ref[2].grad), (bb.grad, ref[3].grad)):
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-5)
    # only the composite weight carries a gradient (the bias output unused)
    Wa.grad = ba.grad = Wb.grad = bb.grad = None
    W, b = Fh.compose_linear(Wa, ba, Wb, bb)
    (W * gW).sum().backward()
    np.testing.assert_allclose(Wb.grad.cpu().numpy(), (gW.double().cpu() @ ref[0].detach().T).numpy(), rtol=1e-5, atol=1e-5)
    assert float(bb.grad.abs().max()) == 0.0


def test_knn_wide_mixed_clouds_flag_handover(dev):
    """knn6w_kernel (24 < k <= 40) hands a cloud to the v5 kernel behind it when one of its workgroups overflows a survivor list or has no
    finite bound; the other clouds of the same call keep its own rows.  One call with all four kinds of cloud: random, N identical points
    (every list overflows), random far from the origin, one with a NaN point -- every row must equal the oracle's."""
    Fh = _fh()
    B, N, C, k = 4, 256, 64, 40
    x = _rand((B, N, C), 515)
    x[1] = 0.25
    x[2] = x[2] * 0.1 + 7.0
    x[3, 100, 5] = float("nan")
    want = knn_canon.knn_point_major(x, k)
    got = Fh.knn_graph(x.view(B * N, C).to(dev), B, N, k).idx.view(B, N, k).cpu().numpy()
    assert got.min() >= 0 and got.max() < N
    assert np.array_equal(got, want), "mismatching rows per cloud: %s" % [(int((got[b] != want[b]).any(-1).sum())) for b in range(B)]
