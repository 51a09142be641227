"""The oracle's two fp8 rules against torch's own e4m3 conversion (CPU): fp8_dequantized_weights (per-output-channel weight store) and
fp8_quantize_rows (per-row activation codes of the W8A8 prefill, include/vlo.h prefill_act_dtype = 1).  Neither is a reference feature
(config 5 exceeds the reference); what is pinned here is that the restated rounding IS OCP e4m3 round-to-nearest-even."""
import torch

from oracle import vlo_oracle as O


def test_e4m3_rne_is_torchs_conversion():
    g = torch.Generator().manual_seed(1)
    x = torch.cat([torch.randn(20000, generator=g) * 100, torch.randn(20000, generator=g) * 0.01, torch.linspace(-448, 448, 4097),
                   torch.tensor([0.0, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10, 447.9, 448.0, -448.0, 0.0009765625 * 1.5])]).clamp(-448, 448)
    assert torch.equal(O.e4m3_rne(x), x.to(torch.float8_e4m3fn).float())


def test_fp8_quantize_rows_rule():
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(64, 512, generator=g) * torch.logspace(-3, 3, 64)[:, None]).bfloat16()
    x[7] = 0
    q, s = O.fp8_quantize_rows(x)
    assert s.shape == (64, 1) and s[7, 0] == 1.0 and not q[7].any()
    amax = x.float().abs().amax(dim=1)
    live = amax > 0
    assert torch.equal(s[live, 0], amax[live] * (1.0 / 448.0))
    assert torch.equal(q, q.to(torch.float8_e4m3fn).float()) and q.abs().max() <= 448
    assert torch.equal(q[live].abs().amax(dim=1), torch.full((int(live.sum()),), 448.0))      # the row maximum lands on the top code
    # dequantised rows sit within half an e4m3 step (2^-4 relative, 2^-10 * scale absolute near zero) of the input
    err = (q * s - x.float()).abs()
    assert (err <= (x.float().abs() * 2.0 ** -4).clamp_min(s * 2.0 ** -10) * 1.0001).all()


def test_act_fp8_forward_changes_only_the_layer_projections():
    spec = O.LLM_SPECS["toy"]
    w = O.fp8_dequantized_weights(O.init_llm_weights(spec, seed=3))
    m = O.LlamaOracle(spec, w, torch.float32)
    x = torch.randn(12, spec.hidden_size, generator=torch.Generator().manual_seed(4))
    a, _ = m.forward(x, None)
    b, _ = m.forward(x, None, act_fp8=True)
    rel = (a - b).abs().max() / a.abs().max()
    assert 0 < rel < 0.25                                    # e4m3 activations move the logits by e4m3-sized steps, not by nothing and not by everything
