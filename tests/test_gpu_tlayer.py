"""Row-complete transformer sub-layer kernels (csrc/tlayer.hip, toist_rowgemm) against fp32 CPU math on the same bf16-rounded operands:
out_proj / linear2 + dropout + residual + LayerNorm in one launch (/root/reference/models/transformer.py:297-303, 376-407), and the
backward direction -- data gradient + residual gradients + LayerNorm backward + the dropout mask of the branch gradient -- incl. the
fold of the attention backward kernel's key-split partial sums.  Tolerances: outputs are bf16 (relative 2^-8) of f32 accumulations."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _bf(t):
    return t.to(BF)


def _close(got, ref, what, rtol=8e-3, atol=2e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    bound = rtol * ref.abs() + atol
    assert bool((err <= bound).all()), f"{what}: max excess {float((err - bound).max()):.3e} (max err {float(err.max()):.3e})"


@pytest.mark.parametrize("M,K", [(800, 256), (3328, 2048), (3333, 256), (37, 768)])
def test_rowgemm_layernorm_forward(dev, M, K):
    from toist_amd import kernels as k
    g = torch.Generator().manual_seed(M + K)
    a = _bf(torch.randn(M, K, generator=g))
    w = _bf(torch.randn(256, K, generator=g) / math.sqrt(K))
    bias = torch.randn(256, generator=g) * 0.1
    res = _bf(torch.randn(M, 256, generator=g))
    add = _bf(torch.randn(M, 256, generator=g))
    gamma, beta = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    ad, wd, resd, addd = a.to(dev), w.to(dev), res.to(dev), add.to(dev)
    z = torch.empty(M, 256, dtype=BF, device=dev)
    y = torch.empty(M, 256, dtype=BF, device=dev)
    y2 = torch.empty(M, 256, dtype=BF, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    k.rowgemm(ad, wd, y, b_kind=k.B_ROWK, epi=k.ROW_LN_FWD, bias=bias.to(dev), res=resd, gamma=gamma.to(dev), beta=beta.to(dev), eps=1e-5,
              z=z, mean=mean, rstd=rstd, add=addd, out2=y2)
    zr = (a.float() @ w.float().t() + bias + res.float())
    _close(z, zr, "z", rtol=4e-3, atol=1e-3)
    zz = z.float().cpu()                                    # statistics of the ROUNDED row, as the backward pass will see it
    mu = zz.mean(1)
    var = ((zz - mu[:, None]) ** 2).mean(1)
    rs = (var + 1e-5).rsqrt()
    torch.testing.assert_close(mean.cpu(), mu, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rstd.cpu(), rs, rtol=1e-4, atol=1e-6)
    yr = (zz - mu[:, None]) * rs[:, None] * gamma + beta
    _close(y, yr, "y", rtol=4e-3, atol=1e-3)
    _close(y2, y.float().cpu() + add.float(), "y + add", rtol=4e-3, atol=1e-3)
    # ---- with dropout: the same mask as the GEMM epilogue + stand-alone LayerNorm path (hash of (seed, m * 256 + n)) ----
    from toist_amd import ops
    seed = 12345
    zu = ops.linear(ad, wd, bias.to(dev), res=resd, drop_where=1, drop_p=0.1, drop_seed=seed)
    yu = torch.empty_like(zu)
    k.layernorm_fwd(zu, gamma.to(dev), beta.to(dev), 1e-5, yu)
    k.rowgemm(ad, wd, y, b_kind=k.B_ROWK, epi=k.ROW_LN_FWD, bias=bias.to(dev), res=resd, gamma=gamma.to(dev), beta=beta.to(dev), eps=1e-5,
              z=z, mean=mean, rstd=rstd, drop_p=0.1, drop_seed=seed)
    dropped = float(((z.float() - resd.float()).abs() < 1e-6).float().mean())
    assert 0.07 < dropped < 0.13, dropped
    _close(z, zu, "z with dropout vs the unfused path", rtol=8e-3, atol=4e-3)
    _close(y, yu, "y with dropout vs the unfused path", rtol=1.6e-2, atol=1e-2)


@pytest.mark.parametrize("M,K,parts", [(800, 256, 4), (3328, 768, 4), (3328, 2048, 1), (53, 768, 3), (800, 3072, 1)])
def test_rowgemm_layernorm_backward(dev, M, K, parts):
    from toist_amd import kernels as k
    g = torch.Generator().manual_seed(M + K + parts)
    w = _bf(torch.randn(K, 256, generator=g) / math.sqrt(K))            # the parameter as stored: [out = K, in = 256] -> data gradient reads it k-major
    res, res2 = _bf(torch.randn(M, 256, generator=g)), _bf(torch.randn(M, 256, generator=g))
    z = _bf(torch.randn(M, 256, generator=g) * 2 + 0.3)
    gamma = torch.rand(256, generator=g) + 0.5
    zz = z.float()
    mu = zz.mean(1)
    rs = (((zz - mu[:, None]) ** 2).mean(1) + 1e-5).rsqrt()
    fold_cols = 256 if parts > 1 else 0
    a_full = torch.randn(M, K, generator=g)
    a = _bf(a_full)
    fold = None
    if parts > 1:
        fold = _bf(torch.randn(parts, M, fold_cols, generator=g) * 0.5)
        a_eff = a.float().clone()
        a_eff[:, :fold_cols] = _bf(fold.float().sum(0)).float()          # summed in f32, rounded once
    else:
        a_eff = a.float()
    ad = a.to(dev).clone()
    dz = torch.empty(M, 256, dtype=BF, device=dev)
    dzd = torch.empty(M, 256, dtype=BF, device=dev)
    dgamma, dbeta = torch.full((256,), 0.25, device=dev), torch.full((256,), -0.5, device=dev)
    seed = 777
    k.rowgemm(ad, w.to(dev), dz, b_kind=k.B_KROW, epi=k.ROW_LN_BWD, res=res.to(dev), res2=res2.to(dev), gamma=gamma.to(dev), z=z.to(dev),
              mean=mu.to(dev), rstd=rs.to(dev), out2=dzd, drop_p=0.1, drop_seed=seed, fold=fold.to(dev) if fold is not None else None,
              fold_cols=fold_cols, dgamma=dgamma, dbeta=dbeta)
    k.flush_reductions()
    gr = a_eff @ w.float() + res.float() + res2.float()
    xh = (zz - mu[:, None]) * rs[:, None]
    gg = gr * gamma
    ref = rs[:, None] * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True))
    _close(dz, ref, "dz", rtol=8e-3, atol=4e-3)
    if parts > 1:       # the folded rows were written back for the weight-gradient GEMM
        assert torch.equal(ad[:, :fold_cols].cpu(), _bf(fold.float().sum(0)))
        assert torch.equal(ad[:, fold_cols:].cpu(), a[:, fold_cols:])
    torch.testing.assert_close(dgamma.cpu(), 0.25 + (gr * xh).sum(0), rtol=2e-3, atol=2e-2 * math.sqrt(M / 800))
    torch.testing.assert_close(dbeta.cpu(), -0.5 + gr.sum(0), rtol=2e-3, atol=2e-2 * math.sqrt(M / 800))
    # the masked copy: same hash as toist_dropout_bf16 on the flat [M, 256] tensor
    want = torch.empty_like(dz)
    k.dropout(dz, 0.1, seed, want)
    keep = want.float() != 0
    frac = float((~keep & (dz.float() != 0)).float().mean())
    assert 0.07 < frac < 0.13, frac
    assert torch.equal((dzd.float() != 0) | (dz.float() == 0), keep | (dz.float() == 0))
    _close(dzd, want, "masked dz", rtol=8e-3, atol=1e-5)


def test_rowgemm_plain(dev):
    from toist_amd import kernels as k
    g = torch.Generator().manual_seed(5)
    M, K = 1000, 768
    a, w = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(K, 256, generator=g) / math.sqrt(K))
    res = _bf(torch.randn(M, 256, generator=g))
    out = torch.empty(M, 512, dtype=BF, device=dev)[:, 256:]                       # a column slice: ldo = 512
    k.rowgemm(a.to(dev), w.to(dev), out, b_kind=k.B_KROW, epi=k.ROW_PLAIN, res=res.to(dev))
    _close(out, a.float() @ w.float() + res.float(), "plain", rtol=4e-3, atol=1e-3)
    wr = _bf(torch.randn(256, K, generator=g) / math.sqrt(K))
    bias = torch.randn(256, generator=g)
    k.rowgemm(a.to(dev), wr.to(dev), out, b_kind=k.B_ROWK, epi=k.ROW_PLAIN, bias=bias.to(dev))
    _close(out, a.float() @ wr.float().t() + bias, "plain forward", rtol=4e-3, atol=1e-3)


def test_fused_layer_programs_match_the_per_op_path(dev):
    """toist_amd.tlayer (LayerNorm fused into the GEMMs on both passes) against the per-op programs of toist_amd.engine / transformer.py on
    the same model and batch, eval mode (no dropout: the two paths draw their seeds in different orders).  The loss is a fixed random
    linear functional of the six layers' logits / boxes (no matcher in between: a flipped near-tied assignment would hide kernel
    errors behind assignment noise): outputs agree to bf16 rounding, every gradient has cosine > 0.997 and norm ratio within 2 %
    (measured: >= 0.998 / 1.4 %; the worst are the FFN weight gradients of the decoder, where a bf16 difference flips ReLU gates)."""
    import toist_amd
    from toist_amd import harness, tlayer
    torch.manual_seed(0)
    args = harness.default_args(device="cuda", contrastive_align_loss=True)
    model, _, _, _ = toist_amd.build_model(args)
    for n, b in model.named_buffers():
        if n.endswith("bn3.weight"):
            b.mul_(0.3)
    model.to(dev).eval()
    samples, tok, _, _ = harness.synthetic_batch(2, 160, 192, tokens=16, seed=6, max_targets=6)
    g = torch.Generator().manual_seed(3)
    w_log, w_box, w_pq = (torch.randn(6, 2, 100, n, generator=g).to(dev) for n in (256, 4, 64))

    def run(flag):
        old = tlayer.ENABLED
        tlayer.ENABLED = flag
        try:
            model.zero_grad(set_to_none=True)
            mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
            out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
            st = out["_stacked"]
            loss = (st["pred_logits"] * w_log).sum() + 30 * (st["pred_boxes"] * w_box).sum() + (st["proj_queries"] * w_pq).sum() + out["proj_tokens"].sum()
            loss.backward()
            torch.cuda.synchronize()
        finally:
            tlayer.ENABLED = old
        grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
        return mc["img_memory"].detach().float().clone(), st["pred_logits"].detach().float().clone(), float(loss), grads

    mem_a, lg_a, loss_a, g_a = run(True)
    mem_b, lg_b, loss_b, g_b = run(False)
    assert float((mem_a - mem_b).norm() / mem_b.norm()) < 1e-2        # six encoder layers, two roundings of the probabilities
    assert float((lg_a - lg_b).norm() / lg_b.norm()) < 1e-2
    assert set(g_a) == set(g_b)
    worst = {}
    for n in g_b:
        if g_b[n].norm() == 0 or "decoder.layers.0.self_attn.in_proj" in n or n.endswith("attention.self.key.bias"):
            continue      # exact zeros: frozen, tgt = 0 in decoder layer 0, a key bias (constant per score row: softmax removes it) -- rounding noise on both sides
        cos = float(torch.nn.functional.cosine_similarity(g_a[n].flatten(), g_b[n].flatten(), dim=0))
        ratio = float(g_a[n].norm() / g_b[n].norm())
        strict = n.startswith(("transformer.", "backbone.", "input_proj."))       # the heads (ReLU MLP on slightly different inputs) get 0.99 / 4 %
        if cos < (0.997 if strict else 0.99) or not ((0.98 < ratio < 1.02) if strict else (0.96 < ratio < 1.04)):
            worst[n] = (round(cos, 5), round(ratio, 4))
    assert not worst, f"{len(worst)} gradients differ between the fused and the per-op programs: {dict(list(worst.items())[:12])}"
