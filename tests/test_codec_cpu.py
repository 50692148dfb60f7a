"""CPU: the EnCodec-decoder oracle (oracle/encodec_oracle.py; parity with the third-party package UNPINNED) -- structural
properties the published architecture guarantees, which the HIP implementation is then held to on the GPU."""
import torch

from oracle import encodec_oracle as eo


def test_decoder_shapes_causality_and_weight_norm():
    sd = eo.make_state_dict(0)
    assert len(sd) == len(eo.state_dict_spec()) == 70
    g = torch.Generator().manual_seed(1)
    codes = torch.randint(0, 1024, (23, 8), generator=g)
    wav = eo.decode(sd, codes)
    assert wav.shape == (23 * eo.HOP,) and torch.isfinite(wav).all() and wav.std() > 1e-3
    # causal model (causal=True: left-only padding, right-trimmed transposed convs, forward LSTM): sample n depends on frames <= n / 320
    c2 = codes.clone()
    c2[15:] = torch.randint(0, 1024, (8, 8), generator=g)
    w2 = eo.decode(sd, c2)
    assert torch.equal(wav[: 15 * eo.HOP], w2[: 15 * eo.HOP]) and not torch.equal(wav[15 * eo.HOP:], w2[15 * eo.HOP:])
    # weight_norm: every output slice of the folded weight has norm g
    w = eo.fold_weight_norm(sd, "decoder.model.0.conv.conv")
    assert torch.allclose(w.flatten(1).norm(dim=1), sd["decoder.model.0.conv.conv.weight_g"].flatten(), rtol=1e-5)
    # shorter than the reflect padding (k - 1 = 6 frames): the zero-extension rule of encodec's pad1d
    tiny = eo.decode(sd, codes[:3])
    assert tiny.shape == (3 * eo.HOP,) and torch.isfinite(tiny).all()
