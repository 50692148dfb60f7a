"""GPU: the HIP EnCodec decoder (valle_amd/csrc/codec.hip through vle_codec_*) against the oracle on synthetic weights in the
encodec package's state-dict naming: fp32 both sides, convolutions as strided-A GEMMs on the fp32 MFMA vs torch's conv ops."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import valle_amd  # noqa: E402
from oracle import encodec_oracle as eo  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def codec():
    sd = eo.make_state_dict(3)
    return sd, valle_amd.EncodecDecoder(sd, device=DEV)


@pytest.mark.parametrize("T", [1, 5, 40, 225])
def test_codec_decode_matches_oracle(codec, T):
    sd, dec = codec
    g = torch.Generator().manual_seed(T)
    codes = torch.randint(0, 1024, (T, 8), generator=g)
    want = eo.decode(sd, codes)
    got = dec.decode_codes(codes.to(DEV)).cpu()
    assert got.shape == want.shape == (T * 320,)
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    assert err <= 2e-4 * max(scale, 1e-3), (err, scale)  # fp32 summation order (MFMA k-chunks vs torch's conv) through ~20 layers + 2T LSTM steps


def test_codec_reference_signature_batch_and_range_check(codec):
    sd, dec = codec
    g = torch.Generator().manual_seed(9)
    codes = torch.randint(0, 1024, (2, 30, 8), generator=g)            # what VALLE.inference returns: (B, T, Q)
    wav = dec.decode([(codes.transpose(2, 1).to(DEV), None)])            # infer.py:261-263
    assert wav.shape == (2, 1, 30 * 320) and wav.device.type == "cuda"
    for b in range(2):
        want = eo.decode(sd, codes[b])
        assert (wav[b, 0].cpu() - want).abs().max().item() <= 2e-4 * want.abs().max().item()
    bad = codes[0].clone()
    bad[3, 2] = 1024
    with pytest.raises(IndexError):
        dec.decode_codes(bad.to(DEV))
    # growing the scratch (longer utterance after a short one) and shrinking back give the same answers
    long_codes = torch.randint(0, 1024, (300, 8), generator=g)
    a = dec.decode_codes(long_codes.to(DEV)).cpu()
    assert (a - eo.decode(sd, long_codes)).abs().max().item() <= 2e-4 * a.abs().max().item()
    b = dec.decode_codes(codes[0].to(DEV)).cpu()
    assert torch.equal(b, wav[0, 0].cpu())


def test_codec_missing_weights_fail_loudly():
    sd = dict(eo.make_state_dict(0))
    del sd["decoder.model.7.block.1.conv.conv.weight_v"]
    with pytest.raises(valle_amd._lib.VleError):
        valle_amd.EncodecDecoder(sd, device=DEV)
