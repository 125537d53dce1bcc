"""GPU: BEV Decoder (ResNet-18 U-Net + heads) on the HIP conv library vs the oracle and vs fixtures from the
reference's Decoder class.  fp32, tolerance 1e-3 max-abs on the logits (north star), typically ~1e-5."""
import pytest
import torch

from util import cases, gold, hashfill, maxabs
from oracle import decoder_ref as DR

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", list(cases.DECODER_CASES))
def test_decoder_forward(tag):
    from streamingflow_amd.models.decoder import Decoder
    G = gold("decoder.npz")
    cin, ncls, npres, nhd, gate, (b, s, h, w) = cases.DECODER_CASES[tag]
    m = Decoder(cin, ncls, npres, nhd, gate).eval()
    sd = cases.decoder_state_dict(m.state_dict())
    m.load_state_dict(sd)
    m = m.cuda()
    x = hashfill.normal("dec_x_" + tag, (b, s, cin, h, w), seed=62)
    out = m(x.cuda())
    with torch.no_grad():
        want = DR.decoder_forward(sd, x, npres)
    worst = 0.0
    for k, v in want.items():
        if v is None:
            assert out[k] is None
            continue
        assert out[k].shape == v.shape, k
        worst = max(worst, maxabs(out[k], v), maxabs(out[k], G[f"{tag}.{k}"]))
    print(tag, "max-abs", worst)
    assert worst <= 1e-3
    with pytest.raises(RuntimeError):
        m.train()(x.cuda())


def test_decoder_200x200_frames_against_oracle():
    """Shipped size: 64 channels, 200 x 200 BEV, 2 frames; the oracle runs the same frames on the host."""
    from streamingflow_amd.models.decoder import Decoder
    cin, ncls, npres, nhd, gate, _ = cases.DECODER_CASES["shipped_gates_small"]
    m = Decoder(cin, ncls, npres, nhd, gate).eval()
    sd = cases.decoder_state_dict(m.state_dict())
    m.load_state_dict(sd)
    m = m.cuda()
    x = hashfill.normal("dec_x_big", (1, 2, cin, 200, 200), seed=63)
    out = m(x.cuda())
    torch.set_num_threads(8)
    with torch.no_grad():
        want = DR.decoder_forward(sd, x, npres)
    for k, v in want.items():
        if v is not None:
            assert maxabs(out[k], v) <= 1e-3, k
