"""Token-selection bookkeeping of ``decoder.generate`` as the reference calls it (src/models/trocr_model.py:306-316:
``max_length=128, num_beams=4, early_stopping=True, pad_token_id, eos_token_id``), separated from the engine so that it can
be pinned on the CPU against HF's own implementation (tests/test_host_cpu.py runs transformers' ``generate`` on a small
RobertaForCausalLM and this module on the same step logits).

``generate`` lives in a third-party dependency (transformers; the reference pins 4.57.0, this image holds 5.15.0), not
under /root/reference; what is restated here is its published algorithm, following transformers/generation/utils.py
``GenerationMixin._beam_search`` (:3208-3508) and its helpers ``_get_top_k_continuations`` (:3077-3129),
``_get_running_beams_for_next_iteration`` (:3131-3151), ``_update_finished_beams`` (:3153-3204),
``_check_early_stop_heuristic`` (:3008-3052), ``_beam_search_has_unfinished_sequences`` (:3055-3075), and the greedy
branch of ``_sample``:

  * every step ranks the 2*num_beams best continuations of [num_beams x vocab] accumulated log-probabilities;
  * a continuation "hits a stopping criterion" when its token is EOS or it reaches max_length; only those among the first
    num_beams ranks may enter the finished list, scored sum-log-prob / generated_length**length_penalty; the finished list
    keeps the best num_beams by score;
  * the running beams of the next step are the best num_beams continuations that did NOT hit a stopping criterion;
  * with early_stopping=True a batch element stops accepting hypotheses once it holds num_beams finished ones, and the
    loop ends when every batch element is full; with early_stopping=False the loop ends when, for every element, the best
    running score / current generated length cannot beat its worst finished score;
  * the result is the best finished hypothesis per element, cropped to the longest one, padded with pad_token_id.

Everything is torch tensor arithmetic on whatever device the logits live on (vectorised over the batch; one host
synchronisation per step for the loop condition).  ``step(t, flat_ids)`` must return the next-token logits [B*beams, V] of
position t given ids[:, :t+1]; ``reorder(flat_rows, t)`` is told which former row each row continues from (KV cache).
"""
from __future__ import annotations

NEG = -1.0e9


def greedy(step, batch: int, max_len: int, pad_id: int, bos_id: int, eos_id: int, device, update=None):
    """``num_beams=1``: argmax per step; finished rows emit padding; stops when every row has emitted EOS.
    ``update(logits, ids, t, done) -> running`` (optional, kzv_greedy_update): the selection in one launch; running[t] = rows
    still running after step t."""
    import torch
    ids = torch.full((batch, max_len), pad_id, dtype=torch.int64, device=device)
    ids[:, 0] = bos_id
    if update is not None:
        # the stop test costs a host round trip: taken every 4th step; the steps run past the end only write padding, and the
        # per-step counters say where the end was
        done8 = torch.zeros(batch, dtype=torch.uint8, device=device)
        running = None
        for t in range(max_len - 1):
            running = update(step(t, ids), ids, t, done8)
            if t % 4 == 3 or t == max_len - 2:
                r = running[:t + 1].tolist()
                if 0 in r:
                    return ids[:, :r.index(0) + 2]
        return ids[:, :max_len]
    done = torch.zeros(batch, dtype=torch.bool, device=device)
    n = 1
    for t in range(max_len - 1):
        nxt = step(t, ids).argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, pad_id), nxt)
        ids[:, t + 1] = nxt
        n = t + 2
        done |= nxt == eos_id
        if bool(done.all()):
            break
    return ids[:, :n]


def beam_search(step, reorder, batch: int, num_beams: int, max_len: int, vocab: int, pad_id: int, bos_id: int, eos_id: int,
                device, early_stopping=True, length_penalty: float = 1.0, topk=None, update=None):
    """``topk(logits [B * nb, vocab] fp32, run_sc [B, nb]) -> (top_lp [B, 2 nb], top_ix [B, 2 nb])``: optional fused ranking
    (kzv_beam_topk on the GPU); None = the torch expression of the same thing below.
    ``update``: optional fused bookkeeping (kzv_beam_update); with it the loop below is replaced by beam_search_fused."""
    import torch
    if update is not None and topk is not None:
        return beam_search_fused(step, reorder, batch, num_beams, max_len, vocab, pad_id, bos_id, eos_id, device, early_stopping,
                                 length_penalty, topk, update)
    B, nb, K = batch, num_beams, 2 * num_beams
    prompt = 1                                                    # decoder prompt = [BOS]
    run_seq = torch.full((B, nb, max_len), pad_id, dtype=torch.int64, device=device)
    run_seq[:, :, 0] = bos_id
    fin_seq = run_seq.clone()
    run_sc = torch.zeros(B, nb, device=device)
    run_sc[:, 1:] = NEG                                           # only beam 0 is live at the first step
    fin_sc = torch.full((B, nb), NEG, device=device)
    fin_done = torch.zeros(B, nb, dtype=torch.bool, device=device)
    fin_len = torch.full((B, nb), prompt, dtype=torch.int64, device=device)
    unsat = torch.ones(B, 1, dtype=torch.bool, device=device)     # "the open beams may still improve the finished ones"
    top_mask = (torch.arange(K, device=device) < nb).view(1, K)
    base = (torch.arange(B, device=device) * nb).view(B, 1)
    cur = prompt
    while True:
        raw = step(cur - 1, run_seq.view(B * nb, max_len))
        if topk is not None:
            top_lp, top_ix = topk(raw, run_sc)
        else:
            logp = torch.log_softmax(raw.float(), dim=-1)
            acc = logp.view(B, nb, vocab) + run_sc.unsqueeze(-1)
        # top K of the nb * vocab continuations in two stages (the K best overall are among each beam's K best): one
        # single-block top-k per beam row instead of a multi-pass radix select over nb * vocab entries per batch element
        if topk is not None:
            pass
        elif vocab > 4 * K:
            blp, bix = acc.topk(K, dim=2)                                  # [B, nb, K]
            top_lp, sel = blp.reshape(B, nb * K).topk(K, dim=1)
            top_ix = (bix + (torch.arange(nb, device=device) * vocab).view(1, nb, 1)).reshape(B, nb * K).gather(1, sel)
        else:
            top_lp, top_ix = acc.view(B, nb * vocab).topk(K, dim=1)
        src, tok = top_ix // vocab, top_ix % vocab
        cand = run_seq.gather(1, src.unsqueeze(-1).expand(B, K, max_len)).clone()
        cand[:, :, cur] = tok
        hits = (tok == eos_id) | (cur + 1 >= max_len)
        # running beams of the next step: the best nb continuations that did not stop
        nxt_ix = (top_lp + hits.float() * NEG).topk(nb, dim=1)[1]
        run_seq = cand.gather(1, nxt_ix.unsqueeze(-1).expand(B, nb, max_len))
        run_sc = (top_lp + hits.float() * NEG).gather(1, nxt_ix)
        rows = (base + src.gather(1, nxt_ix)).reshape(-1)
        # finished list: stopped continuations of rank < nb, normalised by the generated length
        just = hits & top_mask
        fsc = top_lp / float((cur + 1 - prompt) ** length_penalty)
        full = fin_done.all(dim=1, keepdim=True) & (early_stopping is True)
        fsc = fsc + full.float() * NEG + (~unsat).float() * NEG + (~just).float() * NEG
        m_sc = torch.cat((fin_sc, fsc), dim=1)
        m_ix = m_sc.topk(nb, dim=1)[1]
        fin_seq = torch.cat((fin_seq, cand), dim=1).gather(1, m_ix.unsqueeze(-1).expand(B, nb, max_len))
        fin_sc = m_sc.gather(1, m_ix)
        fin_done = torch.cat((fin_done, just), dim=1).gather(1, m_ix)
        fin_len = torch.cat((fin_len, torch.full((B, K), cur + 1, dtype=torch.int64, device=device)), dim=1).gather(1, m_ix)
        cur += 1
        # stop test (one host synchronisation per step)
        best_open = run_sc[:, :1] / float((cur - prompt) ** length_penalty)
        worst_fin = torch.where(fin_done, fin_sc.min(dim=1, keepdim=True)[0], torch.full_like(fin_sc, NEG))
        unsat = unsat & (best_open > worst_fin).any(dim=-1, keepdim=True)
        go_on = unsat.any() & ~(fin_done.all() & (early_stopping is True)) & ~hits.all()
        if not bool(go_on):
            break
        reorder(rows, cur - 1)
    width = int(fin_len[:, 0].max())
    return fin_seq[:, 0, :width]


def beam_search_fused(step, reorder, batch, num_beams, max_len, vocab, pad_id, bos_id, eos_id, device, early_stopping, length_penalty,
                      topk, update, return_state=False):
    """The same loop with the two device kernels: ``topk`` ranks the continuations, ``update(state, top_lp, top_ix, cur) -> (rows,
    flags)`` does everything beam_search does between two steps (state: dict of the tensors below; token rows double-buffered).
    One host synchronisation per step (the three counters of ``flags``)."""
    import torch
    B, nb = batch, num_beams
    st = {"run_seq": [torch.full((B, nb, max_len), pad_id, dtype=torch.int64, device=device) for _ in range(2)],
          "fin_seq": [torch.full((B, nb, max_len), pad_id, dtype=torch.int64, device=device) for _ in range(2)],
          "run_sc": torch.zeros(B, nb, device=device), "fin_sc": torch.full((B, nb), NEG, device=device),
          "fin_done": torch.zeros(B, nb, dtype=torch.uint8, device=device),
          "fin_len": torch.full((B, nb), 1, dtype=torch.int64, device=device),
          "unsat": torch.ones(B, dtype=torch.uint8, device=device), "cur_buf": 0}
    for t in st["run_seq"] + st["fin_seq"]:
        t[:, :, 0] = bos_id
    st["run_sc"][:, 1:] = NEG
    cur = 1
    while True:
        raw = step(cur - 1, st["run_seq"][st["cur_buf"]].view(B * nb, max_len))
        top_lp, top_ix = topk(raw, st["run_sc"])
        rows, flags = update(st, top_lp, top_ix, cur)
        st["cur_buf"] ^= 1
        cur += 1
        # the stop test is taken on the device (flags[3]); the host looks every 4th step -- updates issued past the end are no-ops
        if (cur - 1) % 4 == 0 or cur >= max_len:
            f = flags.tolist()
            if f[3]:
                st["cur_buf"] = f[4] & 1                               # the buffers written by the last update that was applied
                break
        reorder(rows, cur - 1)
    fin_seq = st["fin_seq"][st["cur_buf"]]
    width = int(st["fin_len"][:, 0].max())
    out = fin_seq[:, 0, :width]
    return (out, st) if return_state else out


def make_device_hooks(batch, num_beams, max_len, vocab, eos_id, early_stopping, length_penalty, device):
    """(topk, update) for beam_search on the GPU: kzv_beam_topk and kzv_beam_update through the C ABI, on the current stream,
    writing into buffers allocated here once."""
    import ctypes as C
    import torch
    from . import _lib as L
    lib = L.load()
    B, nb = batch, num_beams
    top_lp = torch.empty(B, 2 * nb, dtype=torch.float32, device=device)
    top_ix = torch.empty(B, 2 * nb, dtype=torch.int64, device=device)
    rows_buf = torch.empty(B * nb, dtype=torch.int64, device=device)
    flags_buf = torch.zeros(5, dtype=torch.int32, device=device)
    keep = {}

    def topk(raw, run_sc):                # log_softmax + beam score + top 2 * nb per image in one launch
        sc = run_sc.contiguous()
        keep["sc"] = sc
        L.check(lib.kzv_beam_topk(raw.data_ptr(), raw.stride(0), sc.data_ptr(), B, nb, vocab, 2 * nb, top_lp.data_ptr(), top_ix.data_ptr(),
                                  L.stream_handle()), "beam_topk")
        return top_lp, top_ix

    def update(st, lp, ix, cur):          # everything between two decoder steps in one launch
        a = st["cur_buf"]
        bs = L.kzv_beam_state(batch=B, num_beams=nb, max_len=max_len, vocab=vocab, eos_id=eos_id,
                              run_seq_in=st["run_seq"][a].data_ptr(), run_seq_out=st["run_seq"][a ^ 1].data_ptr(),
                              fin_seq_in=st["fin_seq"][a].data_ptr(), fin_seq_out=st["fin_seq"][a ^ 1].data_ptr(),
                              run_scores=st["run_sc"].data_ptr(), fin_scores=st["fin_sc"].data_ptr(), fin_done=st["fin_done"].data_ptr(),
                              fin_len=st["fin_len"].data_ptr(), unsatisfied=st["unsat"].data_ptr())
        L.check(lib.kzv_beam_update(C.byref(bs), lp.data_ptr(), ix.data_ptr(), cur, 1 if early_stopping is True else 0, float(length_penalty),
                                    rows_buf.data_ptr(), flags_buf.data_ptr(), L.stream_handle()), "beam_update")
        return rows_buf, flags_buf

    return topk, update


def make_greedy_hook(batch, vocab, pad_id, eos_id, device, max_len=4096):
    """``update`` for greedy() on the GPU (kzv_greedy_update through the C ABI, current stream)."""
    import torch
    from . import _lib as L
    lib = L.load()
    flags = torch.zeros(max(2, int(max_len)), dtype=torch.int32, device=device)     # sequences still running after step t

    def update(logits, ids, t, done8):
        L.check(lib.kzv_greedy_update(logits.data_ptr(), logits.stride(0), vocab, ids.data_ptr(), ids.stride(0), t, done8.data_ptr(), batch,
                                      pad_id, eos_id, flags.data_ptr(), L.stream_handle()), "greedy_update")
        return flags

    return update
