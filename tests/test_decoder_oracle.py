"""CPU: Decoder oracle (oracle/decoder_ref.py) vs fixtures from the reference's Decoder class
(tests/golden/decoder.npz), and state_dict compatibility of the product class."""
import json
import os

import pytest
import torch

from util import GOLD, cases, gold, hashfill, maxabs
from oracle import decoder_ref as DR


@pytest.mark.parametrize("tag", list(cases.DECODER_CASES))
def test_oracle_matches_reference_fixture(tag):
    G = gold("decoder.npz")
    cin, ncls, npres, nhd, gate, (b, s, h, w) = cases.DECODER_CASES[tag]
    keys = json.load(open(os.path.join(GOLD, "decoder_state_dict_keys.json")))[tag]
    sd = cases.decoder_state_dict({k: torch.empty(v) if v else torch.tensor(0) for k, v in keys.items()})
    x = hashfill.normal("dec_x_" + tag, (b, s, cin, h, w), seed=62)
    with torch.no_grad():
        out = DR.decoder_forward(sd, x, npres)
    for k, v in out.items():
        if v is None:
            assert f"{tag}.{k}" not in G
        else:
            assert maxabs(v, G[f"{tag}.{k}"]) <= 1e-6, k


@pytest.mark.parametrize("tag", list(cases.DECODER_CASES))
def test_product_state_dict_matches_reference(tag):
    from streamingflow_amd.models.decoder import Decoder
    cin, ncls, npres, nhd, gate, _ = cases.DECODER_CASES[tag]
    keys = json.load(open(os.path.join(GOLD, "decoder_state_dict_keys.json")))[tag]
    m = Decoder(cin, ncls, npres, nhd, gate)
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert mine == keys
    assert all(float(b.bn2.weight.abs().max()) == 0.0 for b in m.modules() if hasattr(b, "bn2"))   # zero_init_residual
    with pytest.raises(ValueError):
        Decoder(cin, ncls, npres, nhd, dict(gate, predict_instance=False, predict_future_flow=True))
    with pytest.raises(RuntimeError):
        m.eval()(torch.zeros(1, 1, cin, 16, 16))           # CPU tensor: no fallback
