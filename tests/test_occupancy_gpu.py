"""The persistent sweeps beside a kernel that OCCUPIES CUs on another stream.

Both sweeps need a resident block on every one of the 256 CUs (two per CU in the two-group form) and wait for each other
through bounded spins.  On one GPU RCCL's all-reduce is close to a no-op, so nothing so far showed what a collective that
holds CU resources while it waits for a slower peer does to them.  The stand-in (arcvae_debug_occupy: N workgroups of 256-512
threads with tens of KB of LDS, each waiting a few hundred microseconds) is launched on a stream of its own
  * "beside": right behind the step's enqueue, three launches back to back -- they run beside the forward sweep, the seam
    and the BPTT sweep's chunks;
  * "before": in front of the step, so that some CUs are taken when the sweep's blocks arrive;
  * "every CU": one 100-KB workgroup on every CU in front of the step -- no CU can take a sweep block until its stand-in
    has left.
Asserted: no sweep gives up (sync word 500), no gate expires, the forward activations are bit-identical and the gradients
equal to the undisturbed step's within the run-to-run band of the atomic weight-gradient sums (2e-5)."""
import numpy as np
import pytest
import torch

import arcvae_oracle as O
from arcvae_hip._lib import call, stream_ptr
from helpers import DEFAULT, HYPER, build_engine, make_case

pytestmark = pytest.mark.gpu

SCENARIOS = {
    "beside": dict(before=[], after=[(48, 512, 48 * 1024, 300)] * 3),
    "before": dict(before=[(64, 256, 100 * 1024, 500)], after=[]),
    "every CU": dict(before=[(256, 256, 100 * 1024, 250)], after=[(256, 512, 40 * 1024, 200)]),
}


@pytest.mark.parametrize("B,T", [(64, 128), (256, 48)])    # default shape; the two-group forward sweep (two blocks per CU)
@pytest.mark.parametrize("scenario", list(SCENARIOS))
def test_sweeps_survive_a_neighbour_that_occupies_cus(B, T, scenario):
    from arcvae_hip import engine as E
    cfg = DEFAULT
    params, x, cond, eps, coins = make_case(cfg, B, T, 0.9)
    eng, enc, dec = build_engine(cfg, params)
    ws = eng.workspace(B, T)
    assert E.persistent_forward_ok(ws, eng.d)
    names = ["lstm_layer_0.Wh", "lstm_layer_1.Wx", "lstm_layer_0.bias", "embedding.weight", "fc_mu.weight"]

    def step():
        out = eng.train_step(x, cond, eps, coins, lr=2e-4, update=False, **HYPER)
        return out

    for _ in range(3):                                   # undisturbed reference (third step = replayed segments)
        step()
    torch.cuda.synchronize()
    eng.check_gates()
    ref_h = ws.hseq.clone()
    ref_g = {n: enc.g(n).clone() for n in names}
    ref_dec = dec.grad.clone()
    ref_loss = float(ws.scalars[0])

    comm = torch.cuda.Stream()
    sc = SCENARIOS[scenario]
    for rep in range(4):
        with torch.cuda.stream(comm):
            for (blocks, threads, lds, us) in sc["before"]:
                call("arcvae_debug_occupy", blocks, threads, lds, us, stream_ptr())
        step()
        with torch.cuda.stream(comm):
            for (blocks, threads, lds, us) in sc["after"]:
                call("arcvae_debug_occupy", blocks, threads, lds, us, stream_ptr())
        torch.cuda.synchronize()
        assert int(ws.psync[500].item()) == 0, f"{scenario}: a persistent sweep gave up beside the stand-in (rep {rep})"
        eng.check_gates()
        assert torch.equal(ws.hseq, ref_h), f"{scenario}: forward activations changed (rep {rep})"
        assert abs(float(ws.scalars[0]) - ref_loss) <= 1e-6 * max(1.0, abs(ref_loss))
        for n in names:
            dev = float((enc.g(n) - ref_g[n]).abs().max() / ref_g[n].abs().max())
            assert dev < 2e-5, (scenario, rep, n, dev)
        assert float((dec.grad - ref_dec).abs().max() / ref_dec.abs().max()) < 2e-5
