"""Instruction encoder (reference: models/encoders/instruction_encoder.py:11-94):
token embedding (or RxR BERT features) -> packed (bi)directional LSTM/GRU ->
final state, or the zero-padded output sequence.

nn.LSTM / nn.GRU / nn.Embedding are parameter containers only (state_dict
keys `encoder_rnn.weight_ih_l0[_reverse]`, `embedding_layer.weight`).  The
input projection x W_ih^T for ALL steps and both directions is one MFMA GEMM
in time-major layout; the recurrence runs the fused gate kernels step by step
with packed-sequence semantics (steps past a sample's length keep its state
and emit zeros).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class InstructionEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        rnn = nn.GRU if config.rnn_type == "GRU" else nn.LSTM
        self.encoder_rnn = rnn(input_size=config.embedding_size, hidden_size=config.hidden_size,
                               bidirectional=config.bidirectional)
        if config.sensor_uuid == "instruction":
            if config.use_pretrained_embeddings:
                self.embedding_layer = nn.Embedding.from_pretrained(
                    embeddings=self._load_embeddings(), freeze=not config.fine_tune_embeddings)
            else:
                self.embedding_layer = nn.Embedding(num_embeddings=config.vocab_size,
                                                    embedding_dim=config.embedding_size,
                                                    padding_idx=0)

    @property
    def output_size(self):
        return self.config.hidden_size * (1 + int(self.config.bidirectional))

    def _load_embeddings(self):
        import gzip
        import json

        with gzip.open(self.config.embedding_file, "rt") as f:
            return torch.tensor(json.load(f))

    def _direction(self, gi_tm, lengths_ok, active, w_hh, b_hh, reverse):
        """gi_tm [L, B, G*H] time-major input gates of one direction."""
        Lmax, B, _ = gi_tm.shape
        H = self.config.hidden_size
        lstm = self.config.rnn_type == "LSTM"
        h = gi_tm.new_zeros((B, H))
        c = gi_tm.new_zeros((B, H)) if lstm else None
        outs = [None] * Lmax
        steps = range(Lmax - 1, -1, -1) if reverse else range(Lmax)
        for t in steps:
            if lstm:
                h_new, c_new = ops.lstm_cell(gi_tm[t], h, c, w_hh, b_hh)
            else:
                h_new = ops.gru_cell(gi_tm[t], h, w_hh, b_hh)
            if lengths_ok:  # every sample is still running at every step
                h = h_new
                c = c_new if lstm else None
                outs[t] = h_new
            else:
                m = active[t]
                outs[t] = ops.select_rows(m, h_new, None)
                h = ops.select_rows(m, h_new, h)
                if lstm:
                    c = ops.select_rows(m, c_new, c)
        return outs, h

    # Below this a batch is one step of distinct environments (up to 64 envs per GPU in the
    # reference's configs): looking for duplicates there costs three host syncs and finds none.
    # Sequence-mode batches (T*N rows: 5 episodes x ~100 steps, or a DD-PPO minibatch) are larger.
    DEDUP_MIN_ROWS = 128

    def forward(self, observations, distinct=False):
        """`distinct=True` returns (output, inverse): when the batch was encoded once per distinct
        instruction, `output` holds the U distinct rows and `inverse` [B] the row of every batch
        element (None when every row was encoded -- then output is per batch element).

        Sequence-mode batches ([T*N, 200] tokens: a cached-feature DAgger batch,
        dagger_trainer.py:39-114, or a DD-PPO minibatch, rollout_storage.py:154-276) repeat every
        episode's instruction T times (and pad with all-ones rows); the recurrence is run once per
        DISTINCT token row and the result gathered back -- same values, T-fold less LSTM work.
        The gather's autograd adds the T gradients of a row before they enter BPTT."""
        cfg = self.config
        if cfg.sensor_uuid == "instruction":
            tokens = observations["instruction"].long()
            # :79-80 a step counts iff its embedded vector is not all-zero (the token-id count of
            # :72 is overwritten upstream): per-token flag of the table, gathered like the rows
            nonzero_row = (self.embedding_layer.weight.detach() != 0).any(dim=1)
            if tokens.size(0) >= self.DEDUP_MIN_ROWS and os.environ.get("VLNCE_INSTR_DEDUP", "1") != "0":
                uniq, inverse, lengths, lmin, lmax = self._distinct_rows(tokens, nonzero_row)
                if uniq is not None and uniq.size(0) < tokens.size(0):
                    out = self._encode(self._embed_tm(uniq, lmax), (lengths, lmin, lmax),
                                       time_major=True)
                    # [U, C, L] / [U, H] -> rows; the stacked bidirectional final state is [2, U, H]
                    dim = 1 if (cfg.final_state_only and out.dim() == 3) else 0
                    if distinct and dim == 0:
                        return out, inverse
                    out = out.index_select(dim, inverse)
                    return (out, None) if distinct else out
            lengths = nonzero_row[tokens].sum(dim=1)
            if tokens.is_cuda and torch.cuda.is_current_stream_capturing():
                # inside a graph capture (streams.ActGraph): no host sync -- the recurrence runs
                # at the static padded length with the lengths on the device (steps past a row's
                # length keep its state and emit zeros: the same values, more padding)
                out = self._encode(self._embed_tm(tokens, tokens.size(1)),
                                   (lengths, 1, tokens.size(1)), time_major=True)
                return (out, None) if distinct else out
            lmin, lmax = (int(v) for v in torch.stack([lengths.min(), lengths.max()]).tolist())
            if lmin <= 0:
                raise RuntimeError("Length of all samples has to be greater than 0, "
                                   "but found an element in 'lengths' that is <= 0")
            out = self._encode(self._embed_tm(tokens, lmax), (lengths, lmin, lmax), time_major=True)
            return (out, None) if distinct else out
        out = self._encode(observations["rxr_instruction"])
        return (out, None) if distinct else out

    def _embed_tm(self, tokens, lmax):
        """embedding of the first `lmax` tokens of every row, TIME-MAJOR [lmax, B, E]: the rows the
        recurrent layer's input GEMM reads (no [B, 200, E] table lookup followed by a transposed
        copy; the embedding's backward scatters the time-major gradient rows as they are)."""
        tok_tm = tokens[:, :lmax].t().contiguous()
        return ops.embedding(tok_tm, self.embedding_layer.weight, self.embedding_layer.padding_idx)

    @staticmethod
    def _distinct_rows(tokens, nonzero_row):
        """(distinct rows [U, L], inverse [B], their lengths [U], min length, max length) of an
        int64 token matrix, or (None, ...) -- with ONE host sync.  torch.unique(dim=0) sorts whole
        rows (a 2 ms block sort for 500 x 200 tokens) and syncs for its output size; here rows are
        told apart by a 64-bit multiplicative hash, the [B] hash vector is sorted and ranked with
        fixed-size operations, the candidates are verified against the tokens on the device (a
        collision falls back to no de-duplication), and the number of distinct rows, the verdict
        and the length range come back in one transfer."""
        B, L = tokens.shape
        dev = tokens.device
        mult = torch.arange(1, L + 1, device=dev, dtype=torch.int64) * 0x9E3779B97F4A7C15
        key = ((tokens + 0x632BE59BD9B4E019) * mult).sum(dim=1)  # int64 wrap-around arithmetic
        skey, order = key.sort()
        new_group = torch.ones(B, device=dev, dtype=torch.int64)
        new_group[1:] = (skey[1:] != skey[:-1]).to(torch.int64)
        rank = new_group.cumsum(0) - 1                       # group index of every sorted row
        inverse = torch.empty(B, device=dev, dtype=torch.int64)
        inverse[order] = rank
        # first row carrying each key (slots past the number of groups keep B -> clamped: a real row)
        first = torch.full((B,), B, device=dev, dtype=torch.int64)
        first.scatter_reduce_(0, inverse, torch.arange(B, device=dev), reduce="amin")
        cand = tokens.index_select(0, first.clamp_max(B - 1))
        same = (cand.index_select(0, inverse) == tokens).all()
        lengths = nonzero_row[cand].sum(dim=1)
        n_uniq, ok, lmin, lmax = torch.stack(
            [rank[-1] + 1, same.to(torch.int64), lengths.min(), lengths.max()]).tolist()
        if not ok:
            return None, None, None, 0, 0
        return cand[:n_uniq], inverse, lengths[:n_uniq], int(lmin), int(lmax)

    def _encode(self, feats, length_info=None, time_major=False):
        cfg = self.config
        if length_info is None:
            # :79-80 (rxr features) a step counts iff its feature vector is not all-zero;
            # pack_padded_sequence then keeps the first `length` steps of each sample.  One host
            # sync, as upstream (.cpu()).
            lengths = (feats != 0.0).any(dim=2).sum(dim=1)
            if feats.is_cuda and torch.cuda.is_current_stream_capturing():
                lmin, lmax = 1, feats.size(1)  # (see forward: static length inside a capture)
            else:
                lmin, lmax = (int(v) for v in torch.stack([lengths.min(), lengths.max()]).tolist())
        else:
            lengths, lmin, lmax = length_info
        if lmin <= 0:
            raise RuntimeError("Length of all samples has to be greater than 0, "
                               "but found an element in 'lengths' that is <= 0")
        H = cfg.hidden_size
        rnn = self.encoder_rnn
        if time_major:   # feats is already [lmax, B, E] (the token path embeds tokens[:, :lmax].t())
            B, E = feats.size(1), feats.size(2)
            x_tm = feats.reshape(lmax * B, E)
        else:
            B, E = feats.size(0), feats.size(2)
            x_tm = feats[:, :lmax].transpose(0, 1).reshape(lmax * B, E)
        same_len = lmin == lmax
        dirs = [("", False)] + ([("_reverse", True)] if cfg.bidirectional else [])
        kind = 0 if cfg.rnn_type == "LSTM" else 1
        seqs, finals = [], []
        if ops.L().rnn_seq_supported(kind, H) and rnn.bias_hh_l0 is not None:
            # the whole layer -- input projections, the recurrence of both directions in one
            # persistent launch, the outputs in the consumer's [B, L, dirs*H] rows -- as one
            # autograd node whose backward is two library calls (ops.RNNLayerFn)
            need_grad = torch.is_grad_enabled() and (
                x_tm.requires_grad or any(p.requires_grad for p in rnn.parameters()))
            quads = [tuple(getattr(rnn, n + sfx) for n in ("weight_ih_l0", "bias_ih_l0",
                                                           "weight_hh_l0", "bias_hh_l0"))
                     for sfx, _ in dirs]
            seq, finals = ops.rnn_layer(kind, lengths.to(torch.int32).contiguous(), x_tm, B, lmax,
                                        quads, need_grad)
            if cfg.final_state_only:
                return finals[0] if len(finals) == 1 else torch.stack(finals, 0)
            return seq.permute(0, 2, 1)  # logical [B, H*dirs, Lmax]; memory stays [B, L, C]
        active = None
        if not same_len:
            active = (torch.arange(lmax, device=x_tm.device)[:, None] < lengths[None, :]).to(
                torch.uint8).contiguous()
        gis = []
        for sfx, _ in dirs:
            gis.append(ops.linear(x_tm, getattr(rnn, "weight_ih_l0" + sfx),
                                  getattr(rnn, "bias_ih_l0" + sfx)).view(lmax, B, -1))
        for i, (sfx, rev) in enumerate(dirs):
            outs, h_last = self._direction(gis[i], same_len, active,
                                           getattr(rnn, "weight_hh_l0" + sfx),
                                           getattr(rnn, "bias_hh_l0" + sfx), rev)
            finals.append(h_last)
            if not cfg.final_state_only:
                seqs.append(torch.stack(outs, dim=1))  # [B, L, H]
        if cfg.final_state_only:
            # final_state.squeeze(0): [1,B,H] -> [B,H]; a bidirectional [2,B,H] is left as is (App. B-10)
            return finals[0] if len(finals) == 1 else torch.stack(finals, 0)
        seq = seqs[0] if len(seqs) == 1 else torch.cat(seqs, dim=2)
        return seq.permute(0, 2, 1)  # logical [B, H*dirs, Lmax]; memory stays [B, L, C]
