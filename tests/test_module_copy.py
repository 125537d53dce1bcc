"""CPU + GPU: a module that has run keeps process-local things beside its parameters — packed weight copies (ctypes structs full of
device pointers), the folded head pack, captured hipGraphs.  None of it may travel with ``copy.deepcopy(model)`` / ``torch.save(model)``
(users of the reference copy models for EMA / checkpoint whole modules): ``__getstate__`` drops it and the copy rebuilds it lazily."""
import copy
import ctypes
import io
import pickle
import threading

import pytest
import torch

from util import build_pair, hashfill, maxabs


def test_runtime_state_does_not_travel_with_a_copy():
    net, _ = build_pair(8, device="cpu")
    # what a forward on the GPU leaves behind, as unpicklable stand-ins
    net.gru_ode._graphs["k"] = {"exec": ctypes.c_void_p(1), "lock": threading.Lock()}
    net.gru_ode._graph_structures_seen.add("k")
    net.spatial_grus[0].__dict__["_sf_pack"] = (None, ctypes.c_void_p(5), 0)
    net.res_blocks[1].__dict__["_sf_pack"] = (None, threading.Lock(), 0)
    net.gru_ode.gru_c.__dict__["_sf_general"] = (None, threading.Lock())
    net.__dict__["_folded_tail"] = threading.Lock()
    twin = copy.deepcopy(net)
    assert "_folded_tail" not in twin.__dict__
    assert "_sf_pack" not in twin.spatial_grus[0].__dict__ and "_sf_pack" not in twin.res_blocks[1].__dict__
    assert "_sf_general" not in twin.gru_ode.gru_c.__dict__
    assert len(twin.gru_ode._graphs) == 0 and len(twin.gru_ode._graph_structures_seen) == 0
    # the original keeps everything it had
    assert "k" in net.gru_ode._graphs and "_sf_pack" in net.spatial_grus[0].__dict__ and "_folded_tail" in net.__dict__
    a, b = net.state_dict(), twin.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    buf = io.BytesIO()
    torch.save(net, buf)                          # whole-module checkpoint
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert list(back.state_dict()) == list(a) and "_folded_tail" not in back.__dict__
    pickle.dumps(net.res_blocks[0][0])


@pytest.mark.gpu
def test_a_copy_of_a_model_that_has_run_computes_the_same():
    net, _ = build_pair(8)
    T, B, H, W, C = 3, 2, 48, 40, 8
    x = hashfill.normal("copy_head_x", (T, B, H, W, C), 41).cuda()
    with torch.no_grad():
        y0 = net.head_nhwc(x)
        assert "_folded_tail" in net.__dict__ and "_sf_pack" in net.spatial_grus[0].__dict__
        twin = copy.deepcopy(net)
        assert "_folded_tail" not in twin.__dict__
        y1 = twin.head_nhwc(x)
        y2 = net.head_nhwc(x)
    assert torch.equal(y0, y1) and torch.equal(y0, y2)
