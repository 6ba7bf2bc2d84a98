"""Beam search on the device (N1): kzv_beam_topk + kzv_beam_update against kzv/beam.py's torch statement of the same
bookkeeping -- the statement tests/test_host_cpu.py pins against transformers' own generate() -- on synthetic step functions
(logits = a fixed function of the step and the newest token), so that every branch is visited: beams that stop early, finished
lists that fill up, early_stopping on and off, length penalties, the max_length cut."""
import pytest
import torch

from kzv import beam as BM

pytestmark = pytest.mark.gpu
DEV = "cuda"
PAD, BOS, EOS = 1, 0, 2


def _step_fn(V, T, seed, eos_bias):
    g = torch.Generator().manual_seed(seed)
    W = (torch.randn(T, V, V, generator=g) * 2.0).to(DEV)
    W[:, :, EOS] += eos_bias
    W[:, :, PAD] = -30.0
    W[:, :, BOS] = -30.0

    def step(t, ids):
        return W[t][ids[:, t]].contiguous()
    return step


@pytest.mark.parametrize("B,nb,V,T,eos_bias,early,lp", [
    (7, 4, 157, 14, 1.5, True, 1.0), (7, 4, 157, 14, 1.5, False, 1.0), (5, 4, 41, 12, 3.0, True, 1.0), (5, 4, 41, 12, -2.0, False, 1.0),
    (9, 3, 300, 10, 2.0, True, 0.7), (4, 2, 64, 20, 0.5, False, 1.3), (256, 4, 4300, 16, 4.0, True, 1.0), (3, 8, 97, 9, 2.5, True, 1.0)])
def test_fused_beam_search_equals_the_torch_statement(B, nb, V, T, eos_bias, early, lp):
    parents = []

    def reorder(rows, n):
        parents.append(rows.clone())
    want = BM.beam_search(_step_fn(V, T, B + V, eos_bias), reorder, B, nb, T, V, PAD, BOS, EOS, DEV, early_stopping=early, length_penalty=lp)
    n_ref = len(parents)
    ref_parents, parents = parents, []
    topk, update = BM.make_device_hooks(B, nb, T, V, EOS, early, lp, DEV)
    got, st = BM.beam_search_fused(_step_fn(V, T, B + V, eos_bias), reorder, B, nb, T, V, PAD, BOS, EOS, DEV, early, lp, topk, update,
                                   return_state=True)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want.cpu())
    assert n_ref <= len(parents) <= n_ref + 4                      # the host looks at the stop flag every 4th step
    for a, b in zip(parents, ref_parents):
        assert torch.equal(a, b)                                   # the same re-parenting at every step of the search


@pytest.mark.parametrize("B,V,T,eos_bias", [(5, 41, 12, 2.0), (256, 4300, 20, 5.0), (3, 157, 9, -5.0)])
def test_fused_greedy_equals_the_torch_statement(B, V, T, eos_bias):
    want = BM.greedy(_step_fn(V, T, B + V, eos_bias), B, T, PAD, BOS, EOS, DEV)
    got = BM.greedy(_step_fn(V, T, B + V, eos_bias), B, T, PAD, BOS, EOS, DEV, update=BM.make_greedy_hook(B, V, PAD, EOS, DEV, T))
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want.cpu())
    # first maximum on exact ties, like torch.argmax
    flat = torch.zeros(B, V, device=DEV)
    ids = torch.full((B, T), PAD, dtype=torch.int64, device=DEV)
    done = torch.zeros(B, dtype=torch.uint8, device=DEV)
    flat[:, 7] = 1.0; flat[:, 19] = 1.0
    BM.make_greedy_hook(B, V, PAD, EOS, DEV, T)(flat, ids, 0, done)
    torch.cuda.synchronize()
    assert ids[:, 1].tolist() == [7] * B
