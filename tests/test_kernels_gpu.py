"""Per-kernel parity: each C-ABI entry point (through ops.*) against the CPU oracle on seeded inputs."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import stlt_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("M,N,K", [(1, 64, 32), (37, 174, 256), (128, 128, 768), (300, 768, 96), (129, 2304, 768),
                                   (1000, 130, 3072)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear(pkg, M, N, K, act):
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3, scale=0.1)
    y = pkg.ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=act).cpu()
    ref = x.double() @ w.double().t() + b.double()
    if act:
        ref = O.gelu(ref)
    assert y.shape == (M, N)
    assert (y.double() - ref).abs().max().item() <= 2e-5


def test_linear_strided_rows_and_no_bias(pkg):
    # rows of stride 3*K (the CLS-row view used by the last spatial layer)
    M, N, K = 50, 256, 256
    x = _rand(M, 3 * K, seed=4)
    w = _rand(N, K, seed=5, scale=1 / 16)
    y = pkg.ops.linear(x.to(DEV), w.to(DEV), None, rows=M, ldx=3 * K).cpu()
    ref = x[:, :K].double() @ w.double().t()
    assert (y.double() - ref).abs().max().item() <= 2e-5


def test_linear_with_a_contraction_length_that_is_not_a_multiple_of_32(pkg):
    """Round 4: such products run on the fallback kernel (gemm_any.hip) instead of being rejected."""
    x, w = _rand(4, 20, seed=1), _rand(8, 20, seed=2)
    y = pkg.ops.linear(x.to(DEV), w.to(DEV), None).cpu()
    assert (y.double() - x.double() @ w.double().t()).abs().max().item() <= 2e-5


@pytest.mark.parametrize("d,C,with_scores", [(256, 4, False), (768, 4, False), (768, 38, True), (1024, 5, True), (64, 3, True)])
def test_embed(pkg, d, C, with_scores):
    B, T, N = 3, 5, 6
    g = torch.Generator().manual_seed(7)
    cats = torch.randint(0, C, (B, T, N), generator=g)
    boxes = torch.rand(B, T, N, 4, generator=g)
    scores = torch.rand(B, T, N, generator=g)
    sd = {"category_embeddings.weight": _rand(C, d, seed=8), "box_embedding.weight": _rand(d, 4, seed=9, scale=0.5),
          "box_embedding.bias": _rand(d, seed=10, scale=0.5), "score_embeddings.weight": _rand(d, 1, seed=11),
          "score_embeddings.bias": _rand(d, seed=12, scale=0.5), "layer_norm.weight": 1 + _rand(d, seed=13, scale=0.1),
          "layer_norm.bias": _rand(d, seed=14, scale=0.1)}
    batch = {"categories": cats, "boxes": boxes}
    if with_scores:
        batch["scores"] = scores
    ref = O.category_box_embeddings({k: v.double() for k, v in sd.items()}, "", batch, 1e-12)
    D = {k: v.to(DEV) for k, v in sd.items()}
    got = pkg.ops.embed(cats.to(DEV), boxes.to(DEV), scores.to(DEV) if with_scores else None,
                        D["category_embeddings.weight"], D["box_embedding.weight"], D["box_embedding.bias"],
                        D["score_embeddings.weight"], D["score_embeddings.bias"], D["layer_norm.weight"],
                        D["layer_norm.bias"], 1e-12).cpu()
    assert got.shape == (B, T, N, d)
    assert (got.double() - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("d,with_scores", [(64, True), (512, False), (768, True), (1024, True)])
def test_embed_of_many_tokens_is_bit_identical_to_the_per_token_kernel(pkg, d, with_scores):
    """Round 6: from STLT_EMBED_ROWS tokens (32 768) a wave embeds 8 consecutive tokens with the parameters in registers
    (rowwise.hip embed_rows_kernel); below it one token per wave.  Same rows from either, and the oracle's on a sample.
    The token count is not a multiple of the 32 tokens of a workgroup: the last wave holds 5 tokens."""
    C, n = 9, 32768 + 8 * 3 + 5
    g = torch.Generator().manual_seed(21)
    cats = torch.randint(0, C, (1, n, 1), generator=g)
    boxes = torch.rand(1, n, 1, 4, generator=g)
    scores = torch.rand(1, n, 1, generator=g)
    sd = {"category_embeddings.weight": _rand(C, d, seed=8), "box_embedding.weight": _rand(d, 4, seed=9, scale=0.5),
          "box_embedding.bias": _rand(d, seed=10, scale=0.5), "score_embeddings.weight": _rand(d, 1, seed=11),
          "score_embeddings.bias": _rand(d, seed=12, scale=0.5), "layer_norm.weight": 1 + _rand(d, seed=13, scale=0.1),
          "layer_norm.bias": _rand(d, seed=14, scale=0.1)}
    D = {k: v.to(DEV) for k, v in sd.items()}

    def run(lo, hi):
        return pkg.ops.embed(cats[:, lo:hi].contiguous().to(DEV), boxes[:, lo:hi].contiguous().to(DEV),
                             scores[:, lo:hi].contiguous().to(DEV) if with_scores else None, D["category_embeddings.weight"],
                             D["box_embedding.weight"], D["box_embedding.bias"], D["score_embeddings.weight"],
                             D["score_embeddings.bias"], D["layer_norm.weight"], D["layer_norm.bias"], 1e-12)

    whole = run(0, n)
    assert whole.shape == (1, n, 1, d)
    for lo, hi in ((0, 4096), (16384, 16384 + 999), (n - 2048, n)):  # per-token kernel: fewer than 32 768 tokens per call
        assert torch.equal(whole[:, lo:hi], run(lo, hi)), (lo, hi)
    batch = {"categories": cats[:, n - 300:], "boxes": boxes[:, n - 300:]}
    if with_scores:
        batch["scores"] = scores[:, n - 300:]
    ref = O.category_box_embeddings({k: v.double() for k, v in sd.items()}, "", batch, 1e-12)
    assert (whole[:, n - 300:].cpu().double() - ref).abs().max().item() <= 2e-5


def test_dpp_wave_sum_is_the_butterfly_sum_bit_for_bit(tmp_path):
    """Round 6: the LayerNorm reductions of rowwise.hip run on v_permlane32_swap / v_permlane16_swap / DPP instead of six ds_bpermute
    steps (csrc/wave_dpp.h).  tools/wave_sum_check.hip runs both on 2^20 values of mixed magnitude: same bits in every lane."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "wave_sum_check")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-I", os.path.join(root, "revisiting-spatial-temporal-layouts_amd", "csrc"),
                    os.path.join(root, "tools", "wave_sum_check.hip"), "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout.strip())
    assert r.returncode == 0, r.stdout + r.stderr
    assert '"mismatches": 0' in r.stdout


def _attn_ref(qkv, kpm, causal, H):
    S, L, _ = qkv.shape
    masked = kpm[:, None, :].expand(S, L, L).clone()
    if causal:
        masked |= torch.triu(torch.ones(L, L, dtype=torch.bool), diagonal=1)[None]
    m = torch.zeros(S, L, L, dtype=torch.float64).masked_fill(masked, float("-inf"))
    return O.attention_core(qkv.double(), m, H)


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 7, 8, 9, 11, 13, 15, 16, 17, 24, 31, 32, 33, 36, 37, 47, 48, 49, 64, 65, 100, 256])
@pytest.mark.parametrize("causal", [False, True])
def test_attn_core(pkg, L, causal):
    S, H = (9 if L < 100 else 3), 4
    d = 64 * H
    qkv = _rand(S, L, 3 * d, seed=L, scale=1.5)
    g = torch.Generator().manual_seed(100 + L)
    kpm = torch.rand(S, L, generator=g) < 0.3
    kpm[:, 0] = False  # invariant §8b: key 0 is never masked
    got = pkg.ops.attn_core(qkv.to(DEV), kpm.to(DEV), causal, H).cpu()
    ref = _attn_ref(qkv, kpm, causal, H)
    assert torch.isfinite(got).all()
    assert (got.double() - ref).abs().max().item() <= 2e-5


def test_attn_core_peaked_and_fully_masked_rows(pkg):
    # large logits (online-softmax rescale across key tiles) and a sequence whose keys are all masked -> zeros
    S, L, H = 4, 70, 2
    d = 64 * H
    qkv = _rand(S, L, 3 * d, seed=5, scale=6.0)
    kpm = torch.zeros(S, L, dtype=torch.bool)
    kpm[1, :] = True
    kpm[2, :40] = True  # first key tile fully masked for sequence 2
    got = pkg.ops.attn_core(qkv.to(DEV), kpm.to(DEV), False, H).cpu()
    ref = _attn_ref(qkv, kpm, False, H)
    assert torch.isfinite(got).all()
    assert got[1].abs().max().item() == 0.0
    assert (got.double() - ref).abs().max().item() <= 1e-4


@pytest.mark.parametrize("d", [64, 256, 768, 1024, 2048])
@pytest.mark.parametrize("with_res", [True, False])
def test_add_layernorm(pkg, d, with_res):
    M = 37
    x, r = _rand(M, d, seed=1, scale=3), _rand(M, d, seed=2, scale=3)
    w, b = 1 + _rand(d, seed=3, scale=0.2), _rand(d, seed=4, scale=0.2)
    got = pkg.ops.add_layernorm(x.to(DEV), r.to(DEV) if with_res else None, w.to(DEV), b.to(DEV), 1e-5).cpu()
    ref = O.layer_norm((x + r if with_res else x).double(), w.double(), b.double(), 1e-5)
    assert (got.double() - ref).abs().max().item() <= 2e-5


def test_frames_embed_and_gather(pkg):
    B, T, N, d = 3, 6, 4, 256
    sp = _rand(B, T, N, d, seed=1)
    ft = torch.randint(0, 5, (B, T), generator=torch.Generator().manual_seed(2))
    P, F = _rand(256, d, seed=3), _rand(5, d, seed=4)
    w, b = 1 + _rand(d, seed=5, scale=0.1), _rand(d, seed=6, scale=0.1)
    got = pkg.ops.frames_embed(sp.to(DEV), ft.to(DEV), P.to(DEV), F.to(DEV), w.to(DEV), b.to(DEV), 1e-12).cpu()
    ref = O.layer_norm((sp[:, :, 0] + P[:T][None] + F[ft]).double(), w.double(), b.double(), 1e-12)
    assert (got.double() - ref).abs().max().item() <= 2e-5
    lengths = torch.tensor([6, 2, 4])
    h = pkg.ops.gather_last(got.to(DEV), lengths.to(DEV)).cpu()
    assert torch.equal(h, got[torch.arange(B), lengths - 1])  # pure data movement: bit exact


def test_cpu_tensors_are_refused(pkg):
    with pytest.raises(pkg.StltHipError):
        pkg.ops.linear(torch.zeros(4, 32), torch.zeros(8, 32), None)


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (1000, 768, 2304), (57, 3072, 768), (512, 132, 64)])
def test_gemm_nn_dx_layout(pkg, M, N, K):
    """dX = dY·W (+ residual grad): a (M,K) k-contiguous, b stored (K,N) = torch weight (out=K, in=N)."""
    a, b, r = _rand(M, K, seed=1), _rand(K, N, seed=2, scale=1 / math.sqrt(K)), _rand(M, N, seed=3)
    ref = a.double() @ b.double()
    got = pkg.ops.gemm(a.to(DEV), b.to(DEV), trans_b=True).cpu()
    assert (got.double() - ref).abs().max().item() <= 3e-5
    got = pkg.ops.gemm(a.to(DEV), b.to(DEV), trans_b=True, add=r.to(DEV)).cpu()
    assert (got.double() - (ref + r.double())).abs().max().item() <= 3e-5


@pytest.mark.parametrize("M,N,K,split", [(768, 768, 448, 1), (2304, 768, 1024, 4), (200, 132, 96, 3), (3072, 768, 14336, 14),
                                         (176, 768, 64, 1)])
def test_gemm_tn_dw_layout_with_split_k(pkg, M, N, K, split):
    """dW = dY^T·X: both operands contraction-major ((K,M) and (K,N)); split-K slabs + deterministic reduction."""
    a, b = _rand(K, M, seed=4), _rand(K, N, seed=5)
    ref = a.double().t() @ b.double()
    got = pkg.ops.gemm(a.to(DEV), b.to(DEV), trans_a=True, trans_b=True, n_split=split)
    assert (got.cpu().double() - ref).abs().max().item() <= 2e-5 * math.sqrt(K)
    again = pkg.ops.gemm(a.to(DEV), b.to(DEV), trans_a=True, trans_b=True, n_split=split)
    assert torch.equal(got, again)  # slab reduction: bitwise reproducible
    if split > 1:
        acc = _rand(M, N, seed=6)
        got2 = pkg.ops.gemm(a.to(DEV), b.to(DEV), trans_a=True, trans_b=True, n_split=split, add=acc.to(DEV)).cpu()
        assert (got2.double() - (ref + acc.double())).abs().max().item() <= 2e-5 * math.sqrt(K)


@pytest.mark.parametrize("Lq,Lk", [(32, 33), (33, 32), (17, 5), (5, 70), (64, 64)])
def test_attn_cross(pkg, Lq, Lk):
    """Queries and keys from different token spaces (CAF cross-attention), key padding on the key side only."""
    S, H = 5, 4
    d = 64 * H
    q = _rand(S, Lq, d, seed=Lq, scale=1.5)
    kv = _rand(S, Lk, 2 * d, seed=100 + Lk, scale=1.5)
    kpm = torch.rand(S, Lk, generator=torch.Generator().manual_seed(3)) < 0.3
    kpm[:, 0] = False
    for mask in (kpm, None):
        got = pkg.ops.attn_cross(q.to(DEV), kv.to(DEV), None if mask is None else mask.to(DEV), H).cpu()
        k, v = kv[..., :d], kv[..., d:]
        qh = q.double().reshape(S, Lq, H, 64).transpose(1, 2)
        kh = k.double().reshape(S, Lk, H, 64).transpose(1, 2)
        vh = v.double().reshape(S, Lk, H, 64).transpose(1, 2)
        sc = qh @ kh.transpose(-1, -2) / 8.0
        if mask is not None:
            sc = sc.masked_fill(mask[:, None, None, :], float("-inf"))
        ref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(S, Lq, d)
        assert (got.double() - ref).abs().max().item() <= 2e-5


def test_linear_relu(pkg):
    x, w, b = _rand(300, 256, seed=1), _rand(512, 256, seed=2, scale=1 / 16), _rand(512, seed=3, scale=0.1)
    y = pkg.ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=2).cpu()
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    assert (y.double() - ref).abs().max().item() <= 2e-5


# Stream-K: with scratch lent, under-filled launches are cut into equal k-step ranges and partial tiles are summed by
# the fix-up kernel.  Shapes: fewer tiles than CUs, a ragged last round (1.3 rounds), ranges shorter than a tile
# (many partials per tile), ragged M/N edges, and a launch small enough that the grid shrinks below the CU count.
SK_SHAPES = [(2048, 768, 768), (2048, 768, 3072), (2048, 2304, 768), (14336, 768, 768), (64, 768, 768), (1000, 174, 256),
             (300, 130, 96), (5000, 3072, 64), (1, 64, 32),
             # more tiles than CUs: whole-tile rounds + a stream-K tail (1 round + 80 tiles; 5 rounds + 64; ragged edges; a
             # tail too short on its own, so the last whole round joins it)
             (14336, 3072, 768), (14336, 768, 3072), (14000, 776, 768), (16640, 1024, 64)]


@pytest.mark.parametrize("M,N,K", SK_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_stream_k(pkg, M, N, K, act):
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3, scale=0.1)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    plain = pkg.ops.linear(xd, wd, bd, act=act)
    with pkg.ops.gemm_scratch():
        y = pkg.ops.linear(xd, wd, bd, act=act)
        again = pkg.ops.linear(xd, wd, bd, act=act)
        nobias = pkg.ops.linear(xd, wd, None, act=act)
    ref = x.double() @ w.double().t()
    f = {0: lambda t: t, 1: O.gelu, 2: torch.relu}[act]
    assert (y.cpu().double() - f(ref + b.double())).abs().max().item() <= 2e-5
    assert (nobias.cpu().double() - f(ref)).abs().max().item() <= 2e-5
    assert torch.equal(y, again)  # fixed summation order
    assert (y - plain).abs().max().item() <= 1e-5  # same products, different association across the k-range cuts


@pytest.mark.parametrize("M,N,K", [(2048, 768, 3072), (2048, 3072, 768), (300, 200, 96), (57, 3072, 768), (14336, 768, 3072), (14336, 3072, 768)])
def test_gemm_nn_stream_k_with_add_source(pkg, M, N, K):
    a, b, r = _rand(M, K, seed=1), _rand(K, N, seed=2, scale=1 / math.sqrt(K)), _rand(M, N, seed=3)
    ref = a.double() @ b.double()
    with pkg.ops.gemm_scratch():
        got = pkg.ops.gemm(a.to(DEV), b.to(DEV), trans_b=True).cpu()
        got_r = pkg.ops.gemm(a.to(DEV), b.to(DEV), trans_b=True, add=r.to(DEV)).cpu()
    assert (got.double() - ref).abs().max().item() <= 3e-5
    assert (got_r.double() - (ref + r.double())).abs().max().item() <= 3e-5


@pytest.mark.parametrize("M,N,K", [(2048, 768, 3072), (2048, 768, 768), (300, 200, 96), (57, 768, 768), (14336, 768, 768)])
@pytest.mark.parametrize("lend", [False, True])
def test_gemm_nt_forward_layout_with_add_source(pkg, M, N, K, lend):
    """y = x·Wᵀ + r (the residual add of a post-norm layer in the product's epilogue), whole tiles / ragged edges,
    with and without stream-K."""
    x, w, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(M, N, seed=3)
    ref = x.double() @ w.double().t() + r.double()
    if lend:
        with pkg.ops.gemm_scratch():
            got = pkg.ops.gemm(x.to(DEV), w.to(DEV), add=r.to(DEV)).cpu()
    else:
        got = pkg.ops.gemm(x.to(DEV), w.to(DEV), add=r.to(DEV)).cpu()
    assert (got.double() - ref).abs().max().item() <= 3e-5


def test_gemm_tn_stream_k(pkg):
    M, N, K = 768, 768, 2048  # 18 tiles: far fewer than CUs
    a, b = _rand(K, M, seed=4), _rand(K, N, seed=5)
    ref = a.double().t() @ b.double()
    with pkg.ops.gemm_scratch():
        got = pkg.ops.gemm(a.to(DEV), b.to(DEV), trans_a=True, trans_b=True).cpu()
    assert (got.double() - ref).abs().max().item() <= 2e-5 * math.sqrt(K)


def test_gemm_scratch_too_small_is_rejected(pkg):
    lib = pkg._lib.load()
    buf = torch.empty(1024, dtype=torch.uint8, device=DEV)
    assert lib.stlt_gemm_set_scratch(buf.data_ptr(), 1024) != 0
    assert lib.stlt_gemm_set_scratch(None, 0) == 0


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("lens", [[7, 1, 3, 7, 7, 2, 5], [1] * 70, [32, 17, 32, 5], [64, 3, 70, 1, 33], [5], [31, 1, 32, 33, 64, 2, 2, 2]])
def test_attn_ragged_matches_per_segment_softmax(pkg, lens, causal):
    """Ragged K3: segments shorter than, equal to and longer than the 32-row tile, straddling tile boundaries."""
    H = 4
    d = 64 * H
    M = sum(lens)
    qkv = _rand(M, 3 * d, seed=M + int(causal), scale=1.5)
    got = pkg.ops.attn_ragged(qkv.to(DEV), lens, H, causal=causal).cpu()
    ref = torch.zeros(M, d, dtype=torch.float64)
    r0 = 0
    for n in lens:
        q, k, v = [qkv[r0:r0 + n, i * d:(i + 1) * d].double().view(n, H, 64).transpose(0, 1) for i in range(3)]
        sc = q @ k.transpose(1, 2) / 8.0
        if causal:
            sc = sc.masked_fill(torch.ones(n, n, dtype=torch.bool).triu(1), float("-inf"))
        ref[r0:r0 + n] = (torch.softmax(sc, -1) @ v).transpose(0, 1).reshape(n, d)
        r0 += n
    assert (got.double() - ref).abs().max().item() <= 2e-5


# ---- op-level autograd over the per-kernel backward entry points, against torch autograd in float64
def _leaf(t):
    return t.detach().clone().to(DEV).requires_grad_(True)


@pytest.mark.parametrize("M,N,K", [(40, 64, 32), (300, 768, 96), (1000, 174, 256), (2048, 768, 768), (33, 132, 64), (7, 768, 2048)])
def test_linear_autograd(pkg, M, N, K):
    x, w, b, g = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K)), _rand(N, seed=3, scale=0.1), _rand(M, N, seed=4)
    xd, wd, bd = _leaf(x), _leaf(w), _leaf(b)
    pkg.ops.LinearFn.apply(xd, wd, bd).backward(g.to(DEV))
    x64, w64, b64 = [t.double().requires_grad_(True) for t in (x, w, b)]
    (x64 @ w64.t() + b64).backward(g.double())
    for got, ref, name in ((xd.grad, x64.grad, "dx"), (wd.grad, w64.grad, "dw"), (bd.grad, b64.grad, "db")):
        scale = max(ref.abs().max().item(), 1e-6)
        assert (got.cpu().double() - ref).abs().max().item() / scale <= 2e-5, name


@pytest.mark.parametrize("Lq,Lk,causal,packed", [(32, 33, False, False), (33, 32, False, False), (7, 7, False, True), (32, 32, True, True),
                                                  (64, 64, True, True), (5, 61, False, False),
                                                  # above 64 tokens on either side: the streamed backward (query tiles of 32, keys in tiles)
                                                  (100, 100, True, True), (65, 33, False, False), (33, 130, False, False), (256, 256, True, True),
                                                  (70, 70, False, True)])
def test_attention_autograd(pkg, Lq, Lk, causal, packed):
    S, H = 3, 4
    d = 64 * H
    g = _rand(S, Lq, d, seed=9)
    kpm = torch.rand(S, Lk, generator=torch.Generator().manual_seed(3)) < 0.3
    kpm[:, 0] = False
    if packed:  # q, k, v are views of one packed projection
        qkv = _leaf(_rand(S, Lq, 3 * d, seed=5, scale=1.5))
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        leaves = [qkv]
    else:
        qd = _leaf(_rand(S, Lq, d, seed=5, scale=1.5))
        kv = _leaf(_rand(S, Lk, 2 * d, seed=6, scale=1.5))
        q, k, v = qd, kv[..., :d], kv[..., d:]
        leaves = [qd, kv]
    pkg.ops.AttnFn.apply(q, k, v, kpm.to(DEV), causal, H).backward(g.to(DEV))
    ref_leaves = [t.detach().cpu().double().requires_grad_(True) for t in leaves]
    if packed:
        q64, k64, v64 = ref_leaves[0][..., :d], ref_leaves[0][..., d:2 * d], ref_leaves[0][..., 2 * d:]
    else:
        q64, k64, v64 = ref_leaves[0], ref_leaves[1][..., :d], ref_leaves[1][..., d:]
    sp = lambda t, Lx: t.reshape(S, Lx, H, 64).transpose(1, 2)
    sc = sp(q64, Lq) @ sp(k64, Lk).transpose(-1, -2) / 8.0
    sc = sc.masked_fill(kpm[:, None, None, :], float("-inf"))
    if causal:
        sc = sc.masked_fill(torch.ones(Lq, Lk, dtype=torch.bool).triu(1), float("-inf"))
    (torch.softmax(sc, -1) @ sp(v64, Lk)).transpose(1, 2).reshape(S, Lq, d).backward(g.double())
    for got, ref in zip(leaves, ref_leaves):
        scale = max(ref.grad.abs().max().item(), 1e-6)
        assert (got.grad.cpu().double() - ref.grad).abs().max().item() / scale <= 2e-5


@pytest.mark.parametrize("with_res", [True, False])
@pytest.mark.parametrize("d", [768, 256, 1024, 1028, 1536, 2048])
def test_add_layernorm_and_gelu_autograd(pkg, with_res, d):
    """d > 1024 takes the wide-row LayerNorm backward (accumulator rows in LDS: the register form spilled there)."""
    M = 77 if d == 768 else 1300
    x, r, w, b, g = _rand(M, d, seed=1), _rand(M, d, seed=2), 1 + _rand(d, seed=3, scale=0.1), _rand(d, seed=4, scale=0.1), _rand(M, d, seed=5)
    xd, rd, wd, bd = _leaf(x), _leaf(r), _leaf(w), _leaf(b)
    out = pkg.ops.AddLayerNormFn.apply(xd, rd if with_res else None, wd, bd, 1e-5)
    pkg.ops.GeluFn.apply(out).backward(g.to(DEV))
    x64, r64, w64, b64 = [t.double().requires_grad_(True) for t in (x, r, w, b)]
    s = x64 + r64 if with_res else x64
    torch.nn.functional.gelu(torch.nn.functional.layer_norm(s, (d,), w64, b64, 1e-5)).backward(g.double())
    pairs = [(xd.grad, x64.grad), (wd.grad, w64.grad), (bd.grad, b64.grad)] + ([(rd.grad, r64.grad)] if with_res else [])
    for got, ref in pairs:
        scale = max(ref.abs().max().item(), 1e-6)
        assert (got.cpu().double() - ref).abs().max().item() / scale <= 2e-5


@pytest.mark.parametrize("Lq,Lk,packed", [(32, 33, False), (7, 7, True), (32, 32, True)])
def test_attention_autograd_with_probability_dropout(pkg, Lq, Lk, packed):
    """Same seed -> same mask in the forward and in the backward: repeatable, mean-preserving, and the analytic
    gradient agrees with a central finite difference through the masked graph."""
    S, H, p = 4, 4, 0.3
    d = 64 * H
    if packed:
        base = _rand(S, Lq, 3 * d, seed=5, scale=1.2).to(DEV)
        split = lambda t: (t[..., :d], t[..., d:2 * d], t[..., 2 * d:])
    else:
        base = torch.cat([_rand(S, Lq, d, seed=5, scale=1.2).reshape(-1), _rand(S, Lk, 2 * d, seed=6, scale=1.2).reshape(-1)]).to(DEV)
        def split(t):
            q = t[: S * Lq * d].view(S, Lq, d)
            kv = t[S * Lq * d:].view(S, Lk, 2 * d)
            return q, kv[..., :d], kv[..., d:]
    w = _rand(S, Lq, d, seed=8).to(DEV)

    def run(t, seed, prob=p):
        torch.manual_seed(seed)
        pkg.ops.AttnFn._site = 0x100  # the site id advances per call: pin it so that two runs draw the same mask
        q, k, v = split(t)
        return pkg.ops.AttnFn.apply(q, k, v, None, packed and Lq == 32, H, prob)

    x = base.clone().requires_grad_(True)
    out = run(x, 1)
    (out * w).sum().backward()
    again = run(base, 1)
    other = run(base, 2)
    clean = run(base, 1, 0.0)
    assert torch.equal(out.detach(), again) and not torch.equal(again, other)
    assert (out.detach() - clean).abs().max().item() > 1e-3          # the mask really bites
    assert abs(out.detach().mean().item() - clean.mean().item()) < 0.05  # inverted dropout keeps the scale
    direction = torch.randn(base.shape, generator=torch.Generator().manual_seed(3)).to(DEV) * 1e-2
    fd = ((run(base + direction, 1) * w).sum() - (run(base - direction, 1) * w).sum()).item() / 2
    an = (x.grad * direction).sum().item()
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1e-2), (fd, an)


def test_linear_randomised_shapes_plain_and_stream_k(pkg):
    """40 random (M, N, K, activation) problems, each through the plain persistent path and the stream-K path, against fp64."""
    rng = np.random.Generator(np.random.PCG64(2024))
    worst = 0.0
    for i in range(40):
        M = int(rng.integers(1, 6000)) if i % 4 else int(rng.choice([1, 31, 255, 256, 257, 4096]))
        N = int(rng.integers(1, 3200)) if i % 3 else int(rng.choice([1, 127, 128, 129, 174, 2304]))
        K = 32 * int(rng.integers(1, 100 if M * N < 4_000_000 else 24))
        act = int(rng.integers(0, 3))
        x, w, b = _rand(M, K, seed=i), _rand(N, K, seed=100 + i, scale=1 / math.sqrt(K)), _rand(N, seed=200 + i, scale=0.1)
        ref = x.double() @ w.double().t() + b.double()
        ref = {0: lambda t: t, 1: O.gelu, 2: torch.relu}[act](ref)
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
        plain = pkg.ops.linear(xd, wd, bd, act=act).cpu().double()
        with pkg.ops.gemm_scratch():
            sk = pkg.ops.linear(xd, wd, bd, act=act).cpu().double()
        tol = 2e-5 * max(1.0, math.sqrt(K / 768))
        e1, e2 = (plain - ref).abs().max().item(), (sk - ref).abs().max().item()
        worst = max(worst, e1, e2)
        assert e1 <= tol and e2 <= tol, (M, N, K, act, e1, e2)
    print(f"worst abs error over 40 random linear problems: {worst:.2e}")


def test_dropout_op_mask_scale_and_backward(pkg):
    """ops.dropout: counter-based mask with the expected keep rate, survivors scaled by 1/(1-p), the backward applies the
    same mask, identity in eval mode / at p = 0, odd element counts."""
    for n, p in (((257, 33), 0.1), ((1000003,), 0.5), ((4, 768), 0.25)):
        x = _leaf(_rand(*n, seed=2) + 3.0)  # no zeros in the input: a zero in the output is a dropped element
        y = pkg.ops.dropout(x, p, True)
        kept = y != 0
        rate = kept.float().mean().item()
        assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / x.numel()) ** 0.5 + 1e-3, (n, p, rate)
        assert (y[kept] - x.detach()[kept] / (1 - p)).abs().max().item() <= 1e-6 * 8
        g = _rand(*n, seed=3).to(DEV)
        y.backward(g)
        assert (x.grad[kept] - g[kept] / (1 - p)).abs().max().item() <= 1e-6 * 8 and (x.grad[~kept] == 0).all()
    x = _leaf(_rand(5, 7, seed=1))
    assert pkg.ops.dropout(x, 0.3, False) is x and pkg.ops.dropout(x, 0.0, True) is x


@pytest.mark.parametrize("shapes", [
    [(64, 96, 64)],                                                   # one small product: a handful of k-steps per workgroup
    [(2048, 768, 3072), (2048, 3072, 768), (2048, 768, 768), (2048, 2304, 768)],      # an encoder layer's four products (64 clips, temporal)
    [(64, 768, 3072), (64, 3072, 768), (64, 768, 768)],               # a tail layer: few rows
    [(416, 132, 260), (32, 300, 36), (1024, 4, 4), (96, 256, 128), (4096, 260, 516)],  # ragged tiles, different contraction lengths
    [(14336, 768, 768), (14336, 2304, 768)],                          # long contractions: ranges inside one tile
])
def test_weight_grad_group_matches_per_product_sums(pkg, shapes):
    """stlt_weight_grad_group: every g_w += dyᵀ·x of the group in one stream-K launch equals the fp64 products, accumulates
    into what g_w held, is bitwise reproducible, and leaves a NULL item alone."""
    gen = torch.Generator().manual_seed(len(shapes))
    items, refs = [], []
    for i, (rows, n_out, k_in) in enumerate(shapes):
        dy = (torch.rand(rows, n_out, generator=gen) * 2 - 1)
        x = (torch.rand(rows, k_in, generator=gen) * 2 - 1)
        g0 = torch.rand(n_out, k_in, generator=gen)
        refs.append(g0.double() + dy.double().t() @ x.double())
        items.append((dy.to(DEV), x.to(DEV), g0))
    outs = []
    for rep in range(2):
        run = [(dy, x, g0.clone().to(DEV)) for dy, x, g0 in items]
        with pkg.ops.gemm_scratch(DEV):
            pkg.ops.weight_grad_group(run)
        torch.cuda.synchronize()
        outs.append([g.cpu() for _, _, g in run])
    for (rows, n_out, k_in), got, again, ref in zip(shapes, outs[0], outs[1], refs):
        tol = 3e-6 * max(1.0, rows ** 0.5)
        assert (got.double() - ref).abs().max().item() <= tol * 8, (rows, n_out, k_in, (got.double() - ref).abs().max().item())
        assert torch.equal(got, again)
    with pytest.raises(pkg._lib.StltHipError):  # no scratch lent: the grouped launch has nowhere to put its partial tiles
        pkg.ops.weight_grad_group([(items[0][0], items[0][1], items[0][2].clone().to(DEV))])


def test_linear_relu_autograd(pkg):
    """LinearFn with the ReLU in the product's epilogue: forward and all three gradients against torch (fp64)."""
    M, N, K = 300, 96, 64
    x, w, b, g = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.2), _rand(N, seed=3), _rand(M, N, seed=4)
    xd, wd, bd = _leaf(x), _leaf(w), _leaf(b)
    y = pkg.ops.LinearFn.apply(xd, wd, bd, pkg._lib.ACT_RELU)
    y.backward(g.to(DEV))
    x64, w64, b64 = [t.double().requires_grad_(True) for t in (x, w, b)]
    ref = torch.relu(x64 @ w64.t() + b64)
    ref.backward(g.double())
    assert (y.detach().cpu().double() - ref.detach()).abs().max().item() <= 1e-5
    for got, r in ((xd.grad, x64.grad), (wd.grad, w64.grad), (bd.grad, b64.grad)):
        assert (got.cpu().double() - r).abs().max().item() / max(r.abs().max().item(), 1e-6) <= 2e-5


@pytest.mark.parametrize("S,L,causal", [(37, 7, False), (1, 7, False), (53, 36, False), (5, 32, True), (101, 4, False), (29, 48, True), (64, 16, False),
                                        (29, 64, True), (31, 57, False), (300, 36, False), (300, 32, True), (200, 64, True), (200, 50, False),
                                        (3000, 7, False)])
def test_attn_core_short_sequences_many_items_and_masked_rows(pkg, S, L, causal):
    """The 16-row-tile kernel (L <= 64): few items (the launch is cut into (item, query block) units) and many (whole
    items per wave, several per wave at the largest counts); sequence counts that leave the last item partly filled, sequences whose keys are
    all padded (zeros out), every head count parity; against the fp64 oracle."""
    H = 12
    d = 64 * H
    qkv = _rand(S, L, 3 * d, seed=S + L, scale=2.0)
    kpm = torch.rand(S, L, generator=torch.Generator().manual_seed(S)) < 0.35
    kpm[:, 0] = False
    if S > 3:
        kpm[2, :] = True  # a fully padded sequence
    got = pkg.ops.attn_core(qkv.to(DEV), kpm.to(DEV), causal, H).cpu()
    ref = _attn_ref(qkv, kpm, causal, H)
    if S > 3:
        assert got[2].abs().max().item() == 0.0
        ref[2] = 0.0
    assert torch.isfinite(got).all()
    assert (got.double() - ref).abs().max().item() <= 2e-5


SMALL_TILES = [(128, 48), (128, 64), (128, 96), (128, 128), (128, 144), (128, 192), (64, 64), (64, 96), (64, 128), (64, 160), (64, 192), (64, 256),
               (32, 128), (32, 192), (32, 256)]  # csrc/gemm16*.hip: tile rows x columns


@pytest.mark.parametrize("tile_rows,tile_cols", SMALL_TILES)
@pytest.mark.parametrize("M,N,K", [(2048, 768, 768), (2112, 2304, 768), (2048, 3072, 768), (1088, 768, 3072), (1000, 1536, 64), (77, 52, 96),
                                   (1, 4, 64), (33000, 768, 128)])
def test_linear_small_tiles_vs_fp64(pkg, M, N, K, tile_rows, tile_cols):
    """csrc/gemm16.hip: the nn.Linear forward on whole tile_rows x tile_cols tiles (under-filled launches), every tile on ragged
    and exact shapes, with bias / GELU / ReLU / residual, against an fp64 product and against the large-tile kernel."""
    x = _rand(M, K, seed=M + K, scale=1.5)
    w = _rand(N, K, seed=N + 1, scale=2.0 / math.sqrt(K))
    b = _rand(N, seed=N + 2, scale=0.5)
    r = _rand(M, N, seed=7)
    xd, wd, bd, rd = x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
    ref = x.double() @ w.double().t()
    tol = 3e-6 * math.sqrt(K) * max(1.0, ref.abs().max().item())
    for act, bias, res in ((0, bd, None), (1, bd, None), (2, bd, None), (0, None, None), (0, bd, rd)):
        got = pkg.ops.linear_small(xd, wd, bias, tile_cols, act=act, residual=res, tile_rows=tile_rows)
        again = pkg.ops.linear_small(xd, wd, bias, tile_cols, act=act, residual=res, tile_rows=tile_rows)
        want = ref + (b.double() if bias is not None else 0.0)
        if act == 1:
            want = torch.nn.functional.gelu(want)
        elif act == 2:
            want = torch.relu(want)
        if res is not None:
            want = want + r.double()
        assert torch.isfinite(got).all() and torch.equal(got, again)
        assert (got.cpu().double() - want).abs().max().item() <= tol, (act, bias is not None, res is not None)
        if res is None:
            big = pkg.ops.linear(xd, wd, bias, act=act)
            assert (got - big).abs().max().item() <= tol
    with pytest.raises(pkg._lib.StltHipError):
        pkg.ops.linear_small(xd[:, :K - 8].contiguous(), wd[:, :K - 8].contiguous(), bd, tile_cols, tile_rows=tile_rows)  # K % 32 != 0
    with pytest.raises(pkg._lib.StltHipError):
        pkg.ops.linear_small(xd, wd, bd, tile_cols + 16 if (tile_rows, tile_cols + 16) not in SMALL_TILES else 80, tile_rows=tile_rows)  # not a tile of this height


@pytest.mark.parametrize("tile_rows,tile_cols", SMALL_TILES)
@pytest.mark.parametrize("K", [64, 96, 128, 256])
def test_linear_small_tiles_bias_strip_under_load(pkg, tile_rows, tile_cols, K):
    """Many short tiles per workgroup (33 000 rows, 2 - 8 k-steps a tile): every tile's accumulators start from its own bias strip.  The
    strip is DMA'd in front of the tile's first k-step, so the counted wait that publishes that step publishes it too; round 4's first
    form issued it one k-step before it was read, behind loads the wait lets stay in flight, and lost that race once in a test run.
    A stale strip is another column tile's bias: the bias here is large and different per column, so it cannot hide in the tolerance."""
    M, N = 33000, 768
    x = _rand(M, K, seed=K, scale=1.0)
    w = _rand(N, K, seed=K + 1, scale=1.0 / math.sqrt(K))
    b = (torch.arange(N, dtype=torch.float32) - N / 2) * 0.25
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ref = (x.double() @ w.double().t() + b.double()).float().to(DEV)
    first = None
    for rep in range(12):
        got = pkg.ops.linear_small(xd, wd, bd, tile_cols, tile_rows=tile_rows)
        assert (got - ref).abs().max().item() <= 1e-3, rep
        first = got if first is None else first
        assert torch.equal(got, first), rep


@pytest.mark.parametrize("tile_rows,tile_cols", SMALL_TILES)
@pytest.mark.parametrize("M,n_out,k_in", [(2048, 768, 768), (2112, 2304, 768), (2048, 3072, 768), (1088, 768, 3072), (1000, 64, 1536), (77, 96, 52),
                                          (1, 64, 4), (20000, 128, 768)])
def test_input_grad_small_tiles_vs_fp64(pkg, M, n_out, k_in, tile_rows, tile_cols):
    """csrc/gemm16.hip, WKN build: dx = dy·W (+ residual) with the weight read as it lies ([k][n] image gathered in the kernel), every tile
    width, ragged and exact shapes, against an fp64 product and against the large-tile NN kernel."""
    dy = _rand(M, n_out, seed=M + n_out, scale=1.5)
    w = _rand(n_out, k_in, seed=k_in + 1, scale=2.0 / math.sqrt(n_out))
    r = _rand(M, k_in, seed=9)
    dyd, wd, rd = dy.to(DEV), w.to(DEV), r.to(DEV)
    ref = dy.double() @ w.double()
    tol = 3e-6 * math.sqrt(n_out) * max(1.0, ref.abs().max().item())
    got = pkg.ops.input_grad_small(dyd, wd, tile_cols, tile_rows=tile_rows)
    again = pkg.ops.input_grad_small(dyd, wd, tile_cols, tile_rows=tile_rows)
    assert torch.isfinite(got).all() and torch.equal(got, again)
    assert (got.cpu().double() - ref).abs().max().item() <= tol
    got_r = pkg.ops.input_grad_small(dyd, wd, tile_cols, residual=rd, tile_rows=tile_rows)
    assert (got_r.cpu().double() - (ref + r.double())).abs().max().item() <= tol
    if k_in % 4 == 0 and n_out % 32 == 0 and M > 1:
        lib = pkg._lib.load()
        big = torch.empty(M, k_in, device=DEV)
        with pkg.ops.gemm_scratch(DEV):
            pkg._lib.check(lib.stlt_gemm(0, 1, dyd.data_ptr(), n_out, wd.data_ptr(), k_in, None, 0, big.data_ptr(), k_in, 0, M, k_in, n_out, 1,
                                         torch.cuda.current_stream().cuda_stream), "stlt_gemm")
        assert (got - big).abs().max().item() <= tol


def test_linear_dispatch_picks_small_tiles_for_the_under_filled_products(pkg):
    """The launch-time choice (stlt_linear_small_choice): the 2048-row products of the temporal tower at the reference's default batch
    go to whole small tiles, the bench-sized ones stay on the 256 x 128 tiles; and ops.linear gives the small-tile kernel's bits where
    the choice says so."""
    if os.environ.get("STLT_GEMM16") == "0":
        pytest.skip("small-tile routing switched off for this run")
    lib = pkg._lib.load()
    assert lib.stlt_linear_small_choice(2048, 768, 768) > 0 and lib.stlt_linear_small_choice(2048, 3072, 768) > 0
    assert lib.stlt_linear_small_choice(229376, 2304, 768) == 0 and lib.stlt_linear_small_choice(32768, 3072, 768) == 0
    M, N, K = 2048, 768, 768
    tc = lib.stlt_linear_small_choice(M, N, K)
    x, w, b = _rand(M, K, seed=1).to(DEV), _rand(N, K, seed=2, scale=0.05).to(DEV), _rand(N, seed=3).to(DEV)
    with pkg.ops.gemm_scratch(DEV):
        assert torch.equal(pkg.ops.linear(x, w, b, act=1), pkg.ops.linear_small(x, w, b, tc, act=1))


@pytest.mark.parametrize("M", [2048, 2112, 1088, 4096, 300])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_ffn_block_backward_gelu_epilogue_on_small_tiles(pkg, M, p):
    """The FFN hidden gradient du = drop(df·W2) ∘ gelu'(u) with its column sums (lin1_b's gradient) rides in the dX product's epilogue;
    for under-filled launches that product now runs on gemm16.hip's whole small tiles (round 4) instead of stream-K + a fix-up launch.
    Both builds draw the same masks: every gradient of the block must agree between them to rounding, and — without dropout — with
    torch autograd in fp64.  2112 rows: a last 128-row tile that is the first half of a 256-row group (its partner's partial rows of the
    column-sum buffer are zero-filled); 300 rows: three tile rows, the last one ragged."""
    d = 768
    lib = pkg._lib.load()
    if os.environ.get("STLT_GEMM16") != "0":
        assert lib.stlt_input_grad_small_choice(2048, d, 4 * d) > 0 and lib.stlt_input_grad_small_choice(14336, d, 4 * d) == 0
    x, g = _rand(M, d, seed=1), _rand(M, d, seed=2)
    w1, b1 = _rand(4 * d, d, seed=3, scale=1 / math.sqrt(d)), _rand(4 * d, seed=4, scale=0.1)
    w2, b2 = _rand(d, 4 * d, seed=5, scale=1 / math.sqrt(4 * d)), _rand(d, seed=6, scale=0.1)
    ln_w, ln_b = 1 + 0.1 * _rand(d, seed=7), 0.1 * _rand(d, seed=8)
    host = (x, w1, b1, w2, b2, ln_w, ln_b)

    def run(mode):
        pkg.ops.set_gemm_small_tiles(mode)
        try:
            torch.manual_seed(5)
            leaves = [t.clone().to(DEV).requires_grad_(True) for t in host]
            out = pkg.ops.FfnBlockFn.apply(leaves[0], 1e-5, pkg._lib.ACT_GELU, True, p, *leaves[1:])
            out.backward(g.to(DEV))
            return [out.detach()] + [t.grad for t in leaves]
        finally:
            pkg.ops.set_gemm_small_tiles(-2)  # back to the process's setting (a run with STLT_GEMM16=0 must stay on the large tiles)

    small, large = run(-1), run(0)
    names = ("out", "dx", "dw1", "db1", "dw2", "db2", "dln_w", "dln_b")
    for a, b, name in zip(small, large, names):
        scale = max(b.abs().max().item(), 1e-6)
        assert (a - b).abs().max().item() / scale <= 2e-5, name
    assert all(torch.equal(a, b) for a, b in zip(small, run(-1)))  # bitwise reproducible
    for rows in (128, 64, 32):  # the routing restricted to one tile height (round 5: 64- and 32-row tiles): the same gradients from each
        if os.environ.get("STLT_GEMM16") == "0":
            break
        for a, b, name in zip(run(rows), large, names):
            scale = max(b.abs().max().item(), 1e-6)
            assert (a - b).abs().max().item() / scale <= 2e-5, (rows, name)
    if p == 0.0:
        r = [t.double().requires_grad_(True) for t in host]
        hid = torch.nn.functional.gelu(r[0] @ r[1].t() + r[2])
        ref = torch.nn.functional.layer_norm(r[0] + hid @ r[3].t() + r[4], (d,), r[5], r[6], 1e-5)
        ref.backward(g.double())
        for a, want, name in zip(small[1:], r, names[1:]):
            scale = max(want.grad.abs().max().item(), 1e-6)
            assert (a.cpu().double() - want.grad).abs().max().item() / scale <= 1e-4, name


def _mhsa_case(S, L, H, seed):
    d = 64 * H
    x = _rand(S, L, d, seed=seed, scale=1.5)
    w = _rand(3 * d, d, seed=seed + 1, scale=2.0 / math.sqrt(d))
    b = _rand(3 * d, seed=seed + 2, scale=0.5)
    kpm = torch.rand(S, L, generator=torch.Generator().manual_seed(seed)) < 0.3
    kpm[:, 0] = False
    if S > 3:
        kpm[2, :] = True  # a fully padded sequence
    return x, w, b, kpm


@pytest.mark.parametrize("S,H", [(1, 12), (4, 12), (5, 12), (64, 12), (257, 12), (1030, 12), (9, 4), (130, 2)])
def test_mhsa_fused_matches_projection_plus_attention_core(pkg, S, H):
    """stlt_mhsa_fused_fwd: in-projection + causal attention of 32-frame clips in one kernel, against the
    fp64 oracle and against the two-launch path (stlt_linear_fwd + stlt_attn_core_fwd); ragged last clip group, a fully
    padded clip (zeros out), padded frames; sequences of more than 64 tokens are refused."""
    d, L = 64 * H, 32
    x, w, b, kpm = _mhsa_case(S, L, H, S)
    xd, wd, bd, kd = x.to(DEV), w.to(DEV), b.to(DEV), kpm.to(DEV)
    got = pkg.ops.mhsa_fused(xd, wd, bd, kd, H)
    again = pkg.ops.mhsa_fused(xd, wd, bd, kd, H)
    qkv = pkg.ops.linear(xd.view(S * L, d), wd, bd).view(S, L, 3 * d)
    two = pkg.ops.attn_core(qkv, kd, True, H)
    ref = _attn_ref((x.double().view(S * L, d) @ w.double().t() + b.double()).view(S, L, 3 * d), kpm, True, H)
    if S > 3:
        assert got[2].abs().max().item() == 0.0
        ref[2] = 0.0
    assert torch.isfinite(got).all() and torch.equal(got, again)
    assert (got.cpu().double() - ref).abs().max().item() <= 3e-5
    assert (got - two).abs().max().item() <= 2e-5
    with pytest.raises(pkg._lib.StltHipError):
        long = torch.zeros(2, 65, d, device=DEV)
        pkg.ops.mhsa_fused(long, wd, bd, torch.zeros(2, 65, dtype=torch.bool, device=DEV), H)


@pytest.mark.parametrize("L", [1, 2, 5, 7, 8, 13, 16, 17, 18, 24, 31, 32, 33, 34, 36, 40, 47, 48, 49, 50, 57, 63, 64])
@pytest.mark.parametrize("causal", [True, False])
def test_mhsa_fused_every_sequence_length(pkg, L, causal):
    """Round 4: the fused kernel on the reference's real layouts (T = 17 / 33, datasets.py:97-113; N = 5 / 8 objects), cfg4's 64 frames /
    36 objects and everything between: sequences start anywhere inside a 128-row item, the last item is ragged, a sequence is fully
    padded.  Against the fp64 oracle and the two-launch path.  Non-causal sequences above ~36 tokens need more than 5 key blocks per
    query block and are refused (the whole-path code then takes the two launches)."""
    H = 3
    d = 64 * H
    S = max(3, min(41, 700 // L)) + 2
    x, w, b, kpm = _mhsa_case(S, L, H, 1000 + L)
    xd, wd, bd, kd = x.to(DEV), w.to(DEV), b.to(DEV), kpm.to(DEV)
    rows = 128 // L * L

    def key_blocks(window):  # mhsa.hip mhsa16_key_blocks_as: 16-row key blocks from the block holding a sequence's first row, or from that row
        worst, total, inside = 1, 0, True
        for blk in range(0, rows, 16):
            k0 = (blk // L) * L if window else (blk // L) * L // 16 * 16
            last_row = blk + 15 if causal else (min(blk + 15, rows - 1) // L + 1) * L - 1
            n = (last_row - k0) // 16 + 1
            worst, total, inside = max(worst, n), total + n, inside and k0 + 16 * n <= 128
        return worst, total, inside

    worst, total, _ = key_blocks(False)
    if not causal:  # mhsa16_window: the window form where it needs fewer blocks and every block stays inside the 128-row tile
        w_worst, w_total, w_inside = key_blocks(True)
        if w_inside and w_worst <= worst and w_total < total:
            worst = w_worst
    if worst > 5:
        with pytest.raises(pkg._lib.StltHipError):
            pkg.ops.mhsa_fused(xd, wd, bd, kd, H, causal=causal)
        return
    got = pkg.ops.mhsa_fused(xd, wd, bd, kd, H, causal=causal)
    again = pkg.ops.mhsa_fused(xd, wd, bd, kd, H, causal=causal)
    qkv = pkg.ops.linear(xd.view(S * L, d), wd, bd).view(S, L, 3 * d)
    two = pkg.ops.attn_core(qkv, kd, causal, H)
    ref = _attn_ref((x.double().view(S * L, d) @ w.double().t() + b.double()).view(S, L, 3 * d), kpm, causal, H)
    assert got[2].abs().max().item() == 0.0
    ref[2] = 0.0
    assert torch.isfinite(got).all() and torch.equal(got, again)
    assert (got.cpu().double() - ref).abs().max().item() <= 3e-5
    assert (got - two).abs().max().item() <= 2e-5


@pytest.mark.parametrize("L,causal", [(17, True), (32, True), (33, True), (64, True), (50, True), (7, False), (5, False), (8, False), (36, False)])
@pytest.mark.parametrize("p", [0.0, 0.1, 0.5])
def test_mhsa_fused_training_form_vs_masked_oracle(pkg, L, causal, p):
    """The TRAIN build of the fused kernel (what stlt_train_forward launches): packed q / k / v written for the tape equal the
    in-projection product, and the context equals the oracle's attention with the SAME dropout mask on the probabilities (element
    index ((query token * H + head) << 8) | key position: the mask stlt_attn_core_bwd regenerates); also bit-identical to the two-launch
    training path's tape, and the backward kernel accepts the tape (gradient vs fp64 autograd)."""
    H = 2
    d = 64 * H
    S = max(3, 300 // L) + 2
    x, w, b, kpm = _mhsa_case(S, L, H, 2000 + L)
    xd, wd, bd, kd = x.to(DEV), w.to(DEV), b.to(DEV), kpm.to(DEV)
    seed, site = 777 + L, 8 * 5
    ctx, qkv = pkg.ops.mhsa_fused(xd, wd, bd, kd, H, causal=causal, want_qkv=True, dropout_p=p, seed=seed, site=site)
    ctx2, qkv2 = pkg.ops.mhsa_fused(xd, wd, bd, kd, H, causal=causal, want_qkv=True, dropout_p=p, seed=seed, site=site)
    assert torch.equal(ctx, ctx2) and torch.equal(qkv, qkv2)
    qkv_ref = (x.double().view(S * L, d) @ w.double().t() + b.double()).view(S, L, 3 * d)
    assert (qkv.cpu().double() - qkv_ref).abs().max().item() <= 3e-5
    xq = qkv_ref.clone().requires_grad_(True)
    sp = lambda t: t.reshape(S, L, H, 64).transpose(1, 2)
    sc = sp(xq[..., :d]) @ sp(xq[..., d:2 * d]).transpose(-1, -2) / 8.0
    masked = kpm[:, None, None, :].expand(S, H, L, L).clone()
    if causal:
        masked |= torch.ones(L, L, dtype=torch.bool).triu(1)
    pr = torch.nan_to_num(torch.softmax(sc.masked_fill(masked, float("-inf")), -1), nan=0.0)
    if p > 0:
        pr = pr * _drop_mask(O, p, seed, site, S, H, L)
    ref = (pr @ sp(xq[..., 2 * d:])).transpose(1, 2).reshape(S, L, d)
    assert torch.isfinite(ctx).all()
    assert (ctx.cpu().double() - ref.detach()).abs().max().item() <= 5e-5
    assert ctx[2].abs().max().item() == 0.0
    # the reverse sweep's attention backward on this tape
    g = _rand(S, L, d, seed=L + 7)
    dqkv = pkg.ops.attn_core_bwd(qkv, g.to(DEV), kd, causal, H, p, seed, site)
    ref.backward(g.double())
    scale = max(xq.grad.abs().max().item(), 1e-6)
    assert (dqkv.cpu().double() - xq.grad).abs().max().item() / scale <= 5e-5


@pytest.mark.parametrize("L", [17, 24, 31, 32, 33, 36, 40, 47, 48, 49, 57, 63, 64])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("p", [0.0, 0.25])
def test_attn_core_bwd_mfma_blocks_vs_fp64(pkg, L, causal, p):
    """stlt_attn_core_bwd on 2 / 3 / 4 sixteen-row blocks (17-32, 33-48, 49-64 tokens: cfg2's clips, the fusion models' 33
    appearance tokens, cfg4's 36 objects and 64 frames): dqkv and its column sums against torch autograd in fp64 on the oracle's
    attention, with padded keys, a fully padded sequence, and the dropout mask the forward would have drawn."""
    S, H = 21, 3
    d = 64 * H
    qkv = _rand(S, L, 3 * d, seed=L, scale=1.5)
    g = _rand(S, L, d, seed=L + 1)
    kpm = torch.rand(S, L, generator=torch.Generator().manual_seed(L)) < 0.3
    kpm[:, 0] = False
    kpm[4, :] = True  # a fully padded sequence: zero output, zero gradient
    seed, site = 1234 + L, 24
    dqkv, gb = pkg.ops.attn_core_bwd(qkv.to(DEV), g.to(DEV), kpm.to(DEV), causal, H, p, seed, site, want_bias_grad=True)
    x = qkv.double().requires_grad_(True)
    sp = lambda t: t.reshape(S, L, H, 64).transpose(1, 2)
    sc = sp(x[..., :d]) @ sp(x[..., d:2 * d]).transpose(-1, -2) / 8.0
    masked = kpm[:, None, None, :].expand(S, H, L, L).clone()
    if causal:
        masked |= torch.ones(L, L, dtype=torch.bool).triu(1)
    pr = torch.softmax(sc.masked_fill(masked, float("-inf")), -1)
    pr = torch.nan_to_num(pr, nan=0.0)  # fully masked rows
    if p > 0:
        pr = pr * _drop_mask(O, p, seed, site, S, H, L)
    (pr @ sp(x[..., 2 * d:])).transpose(1, 2).reshape(S, L, d).backward(g.double())
    ref = x.grad
    scale = max(ref.abs().max().item(), 1e-6)
    assert torch.isfinite(dqkv).all()
    assert (dqkv.cpu().double() - ref).abs().max().item() / scale <= 2e-5
    assert (gb.cpu().double() - ref.reshape(-1, 3 * d).sum(0)).abs().max().item() / max(ref.reshape(-1, 3 * d).sum(0).abs().max().item(), 1e-6) <= 5e-5
    assert dqkv[4].abs().max().item() == 0.0


@pytest.mark.parametrize("Lq,Lk", [(32, 33), (33, 32), (16, 33), (33, 16), (7, 20), (48, 48), (1, 5), (40, 9)])
@pytest.mark.parametrize("p", [0.0, 0.25])
def test_attn_cross_bwd_mfma_vs_fp64(pkg, Lq, Lk, p):
    """stlt_attn_bwd with queries and keys from different buffers (the fusion models' cross-attention: 32 frames x 33 appearance tokens,
    models.py:411-419) on attn_bwdx16.hip's 16-row blocks — every (query blocks, key blocks) pair of 1 - 3 — against torch autograd in
    fp64: k | v as the two halves of one packed projection (the block path's layout), padded keys, a fully padded sequence, and the
    dropout mask the forward drew."""
    S, H = 13, 3
    d = 64 * H
    q = _rand(S, Lq, d, seed=Lq, scale=1.5)
    kv = _rand(S, Lk, 2 * d, seed=Lk + 100, scale=1.5)
    g = _rand(S, Lq, d, seed=Lq + Lk)
    kpm = torch.rand(S, Lk, generator=torch.Generator().manual_seed(Lk)) < 0.3
    kpm[:, 0] = False
    kpm[4, :] = True  # a fully padded sequence: zero output, zero gradient
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())  # what AttnFn.forward will draw
    torch.manual_seed(77)
    qd = q.to(DEV).requires_grad_(True)
    kvd = kv.to(DEV).requires_grad_(True)
    ctx = pkg.ops.AttnFn.apply(qd, kvd[..., :d], kvd[..., d:], kpm.to(DEV), False, H, p)
    ctx.backward(g.to(DEV))
    qr, kvr = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    sp = lambda t, L_: t.reshape(S, L_, H, 64).transpose(1, 2)
    sc = sp(qr, Lq) @ sp(kvr[..., :d], Lk).transpose(-1, -2) / 8.0
    pr = torch.nan_to_num(torch.softmax(sc.masked_fill(kpm[:, None, None, :].expand(S, H, Lq, Lk), float("-inf")), -1), nan=0.0)
    if p > 0:
        s_, h_, i_, j_ = np.meshgrid(np.arange(S, dtype=np.uint64), np.arange(H, dtype=np.uint64), np.arange(Lq, dtype=np.uint64),
                                     np.arange(Lk, dtype=np.uint64), indexing="ij")
        idx = ((((s_ * np.uint64(Lq) + i_) * np.uint64(H)) + h_) << np.uint64(8)) | j_
        keep = O.dropout_keep(p, seed, pkg.ops.AttnFn._site, idx)
        pr = pr * (torch.from_numpy(keep).double() * float(np.float32(1.0) / (np.float32(1.0) - np.float32(p))))
    ref = (pr @ sp(kvr[..., d:], Lk)).transpose(1, 2).reshape(S, Lq, d)
    assert (ctx.detach().cpu().double() - ref.detach()).abs().max().item() <= 5e-5
    ref.backward(g.double())
    for got, want, name in ((qd.grad, qr.grad, "dq"), (kvd.grad, kvr.grad, "dkv")):
        assert torch.isfinite(got).all(), name
        assert (got.cpu().double() - want).abs().max().item() / max(want.abs().max().item(), 1e-6) <= 2e-5, name
    assert qd.grad[4].abs().max().item() == 0.0 and kvd.grad[4].abs().max().item() == 0.0


def _drop_mask(O, p, seed, site, S, H, L):
    """(S,H,L,L) multiplicative mask of the attention probabilities: element (s,h,i,j) has idx = (((s*L+i)*H + h) << 8) | j."""
    s_, h_, i_, j_ = np.meshgrid(np.arange(S, dtype=np.uint64), np.arange(H, dtype=np.uint64), np.arange(L, dtype=np.uint64),
                                 np.arange(L, dtype=np.uint64), indexing="ij")
    idx = ((((s_ * np.uint64(L) + i_) * np.uint64(H)) + h_) << np.uint64(8)) | j_
    keep = O.dropout_keep(p, seed, site, idx)
    return torch.from_numpy(keep).double() * float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


@pytest.mark.parametrize("M", [1, 17, 64, 65, 128])
@pytest.mark.parametrize("N,K", [(768, 768), (3072, 768), (768, 3072), (174, 768), (100, 200)])
def test_skinny_products_split_k_partial_tiles_match_fp64(pkg, M, N, K):
    """Round 6 (csrc/gemm_any.hip: launch_gemm_skinny): products of at most 128 rows — the one-row-per-clip tail of a tower at a 64-clip batch,
    the prediction head's 174 columns — run as split-k partial tiles + a finishing launch when stream-K scratch is lent: forward with bias /
    GELU / ReLU / add-source, and the input-gradient layout.  Bitwise reproducible."""
    g = torch.Generator().manual_seed(M * 1000 + N + K)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    lib = pkg._lib.load()
    pkg.ops.prof_enable(True)
    try:
        with pkg.ops.gemm_scratch():
            pkg.ops.prof_launches()
            y0 = pkg.ops.linear(x, w, b)
            y0_again = pkg.ops.linear(x, w, b)
            yg = pkg.ops.linear(x, w, b, act=pkg._lib.ACT_GELU)
            yr = pkg.ops.linear(x, w, None, act=pkg._lib.ACT_RELU)
            dy = torch.randn(M, N, generator=g).to(DEV)
            dx = pkg.ops.gemm(dy, w, trans_b=True)  # (M, K) = dy (M, N) · w (N, K): the input-gradient layout
            torch.cuda.synchronize()
            notes = [r["note"] for r in pkg.ops.prof_launches()]
    finally:
        pkg.ops.prof_enable(False)
    assert all("skinny" in n for n in notes) and len(notes) == 5, notes  # every one of them took the split-k path
    plain = pkg.ops.linear(x, w, b)  # outside the context manager `linear` lends the per-stream scratch itself for so few rows: the same path
    ref = x.double() @ w.double().t() + b.double()
    tol = 2e-6 * math.sqrt(K) * max(1.0, ref.abs().max().item())
    assert torch.equal(y0, y0_again) and torch.equal(plain, y0)
    assert (y0.double() - ref).abs().max().item() <= tol and (plain.double() - ref).abs().max().item() <= tol
    assert (yg.double() - torch.nn.functional.gelu(ref)).abs().max().item() <= tol
    assert (yr.double() - torch.relu(x.double() @ w.double().t())).abs().max().item() <= tol
    assert (dx.double() - dy.double() @ w.double()).abs().max().item() <= 2e-6 * math.sqrt(N) * max(1.0, (dy.double() @ w.double()).abs().max().item())
