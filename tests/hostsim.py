"""TEST INFRASTRUCTURE ONLY.  A tensor-level simulator of the C ABI
(include/vlnce_hip.h) written with torch CPU ops, installed in place of
vlnce_amd._lib.HipLib by the `hostsim` fixture so the *host-side* logic of the
package (module wiring, layouts, weight packing, autograd pairing, sequence
handling) can be exercised in the GPU-less container against the golden
vectors.  It never ships in the product path: the package itself has no CPU
fallback and raises when libvlnce_hip.so or a GPU is missing.

Every method documents the contract of the kernel it stands in for; the GPU
tests check the real kernels against the same contracts.
"""
import math

import torch
import torch.nn.functional as F


def _act(v, act):
    if act == 1:
        return torch.relu(v)
    if act == 2:
        return torch.sigmoid(v)
    if act == 3:
        return torch.tanh(v)
    return v


def _mat(t, rows, cols, ld):
    return t.as_strided((rows, cols), (ld, 1))


class HostSim:
    name = "hostsim"
    TILE_ROWS = 128

    # ---- conv / gemm
    def plane_format(self, options=None):
        return 2   # the product default (option "conv_math"); the simulator's arithmetic is fp32

    def conv2d_tiles(self, g):
        M = g["N"] * g["Ho"] * g["Wo"]
        return (M + self.TILE_ROWS - 1) // self.TILE_ROWS, self.TILE_ROWS

    def conv2d_bn_workspace_bytes(self, g):
        return 16

    def bn_finalize_sums(self, acc, M, gamma, beta, eps, momentum, running_mean, running_var,
                         scale_out, mean_out, shift_out=None, rstd_out=None):
        """vlnce_bn_finalize_sums: all copies of the sums -> pending normalisation, running
        statistics like torch (momentum, unbiased variance); acc is zero afterwards."""
        S, Q = acc[:, :, 0].sum(0), acc[:, :, 1].sum(0)
        acc.zero_()
        mean = S / M
        m2 = (Q - S * mean).clamp_min(0.0)
        rstd = 1.0 / torch.sqrt(m2 / M + eps)
        g = gamma.double() if gamma is not None else 1.0
        scale_out.copy_((g * rstd).float())
        mean_out.copy_(mean.float())
        if shift_out is not None:
            b = beta.double() if beta is not None else 0.0
            shift_out.copy_((b - mean * g * rstd).float())
        if rstd_out is not None:
            rstd_out.copy_(rstd.float())
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * mean.float())
            running_var.mul_(1 - momentum).add_(momentum * (m2 / max(M - 1, 1)).float())

    def conv2d_fwd(self, x, w, y, g, in_scale=None, in_shift=None, in_relu=0, scale=None,
                   shift=None, residual=None, ldr=0, act=0, accumulate=0, stat_partial=None,
                   in_center=None, x2=None, in2_scale=None, in2_shift=None, in2_center=None,
                   side_out=None, w_split=None, w_frag=None, bn=None, options=None, w_format=1):
        N, H, W, Cin, Cout = g["N"], g["H"], g["W"], g["Cin"], g["Cout"]
        xi = x.as_strided((N, H, W, Cin), (H * W * g["ldx"], W * g["ldx"], g["ldx"], 1))
        if in_scale is not None:
            xi = (xi - in_center if in_center is not None else xi) * in_scale + in_shift
            if x2 is not None:
                assert g["KH"] == 1 and g["stride"] == 1 and g["pad"] == 0
                x2i = x2.reshape(N, H, W, Cin)
                if in2_scale is not None:
                    x2i = (x2i - in2_center if in2_center is not None else x2i) * in2_scale \
                        + in2_shift
                xi = xi + x2i
            if in_relu:
                xi = torch.relu(xi)
            if side_out is not None:
                side_out.view(N, H, W, Cin).copy_(xi)
        wk = w.view(Cout, g["KH"], g["KW"], Cin).permute(0, 3, 1, 2)
        raw = F.conv2d(xi.permute(0, 3, 1, 2), wk, stride=g["stride"], padding=g["pad"])
        raw = raw.permute(0, 2, 3, 1).reshape(-1, Cout)
        M = raw.size(0)
        if bn is not None:
            # vlnce_bn_sums: {sum x, sum x^2} of the raw output per channel, ADDED to acc (copy 0)
            acc = bn[0]
            acc[0, :, 0] += raw.double().sum(0)
            acc[0, :, 1] += (raw.double() ** 2).sum(0)
        if stat_partial is not None:
            tm, tr = self.conv2d_tiles(g)
            for t in range(tm):
                blk = raw[t * tr:(t + 1) * tr]
                stat_partial[t, :, 0] = blk.sum(0)
                stat_partial[t, :, 1] = ((blk - blk.mean(0)) ** 2).sum(0)
        v = raw
        if scale is not None:
            v = v * scale
        if shift is not None:
            v = v + shift
        if residual is not None:
            v = v + _mat(residual, M, Cout, ldr)
        v = _act(v, act)
        out = _mat(y, M, Cout, g["ldy"])
        if accumulate:
            out += v
        else:
            out.copy_(v)

    def gemm(self, A, lda, transA, B, ldb, transB, Cm, ldc, M, N, K, scale=None, shift=None,
             residual=None, ldr=0, act=0, accumulate=0):
        a = _mat(A, K, M, lda).t() if transA else _mat(A, M, K, lda)
        b = _mat(B, K, N, ldb) if transB else _mat(B, N, K, ldb).t()
        v = a @ b
        if scale is not None:
            v = v * scale
        if shift is not None:
            v = v + shift
        if residual is not None:
            v = v + _mat(residual, M, N, ldr)
        v = _act(v, act)
        out = _mat(Cm, M, N, ldc)
        if accumulate:
            out += v
        else:
            out.copy_(v)

    def colsum(self, x, ldx, M, N, out, accumulate=0):
        s = _mat(x, M, N, ldx).sum(0)
        if accumulate:
            out += s
        else:
            out.copy_(s)

    # ---- norms
    def bn_finalize_workspace_bytes(self, tiles_m, Cc):
        return 0

    def bn_finalize(self, partial, tiles_m, tile_rows, M, Cc, gamma, beta, eps, momentum,
                    running_mean, running_var, scale_out, shift_out, mean_out=None, rstd_out=None,
                    workspace=None):
        p = partial.double()
        n_t = torch.full((tiles_m,), float(tile_rows), dtype=torch.float64)
        n_t[-1] = M - tile_rows * (tiles_m - 1)
        mean = p[:, :, 0].sum(0) / M
        m2 = (p[:, :, 1] + n_t[:, None] * (p[:, :, 0] / n_t[:, None] - mean) ** 2).sum(0)
        var = m2 / M
        rstd = 1.0 / torch.sqrt(var + eps)
        sc = (gamma.double() if gamma is not None else 1.0) * rstd.float().double()
        scale_out.copy_(sc.float())
        shift_out.copy_(((beta if beta is not None else 0.0) - mean.float() * scale_out))
        if mean_out is not None:
            mean_out.copy_(mean.float())
        if rstd_out is not None:
            rstd_out.copy_(rstd.float())
        if running_mean is not None:
            unb = m2 / max(M - 1, 1)
            running_mean.mul_(1 - momentum).add_(momentum * mean.float())
            running_var.mul_(1 - momentum).add_(momentum * unb.float())

    def scale_shift_act(self, x, scale, shift, rows_per_sample, residual, y, M, Cc, act,
                        center=None):
        v = x.reshape(M, Cc)
        if rows_per_sample > 0:
            S = M // rows_per_sample
            v = v.view(S, rows_per_sample, Cc)
            if center is not None:
                v = v - center.view(S, 1, Cc)
            v = (v * scale.view(S, 1, Cc) + shift.view(S, 1, Cc)).reshape(M, Cc)
        else:
            v = (v - center if center is not None else v) * scale + shift
        if residual is not None:
            v = v + residual.reshape(M, Cc)
        y.view(M, Cc).copy_(_act(v, act))

    def gn_chunks(self, HW):
        return (HW + 127) // 128

    def gn_partial(self, x, Nimg, HW, Cc, partial):
        xv = x.reshape(Nimg, HW, Cc)
        for c in range(self.gn_chunks(HW)):
            blk = xv[:, c * 128:(c + 1) * 128]
            partial[:, c, :, 0] = blk.sum(1)
            partial[:, c, :, 1] = (blk * blk).sum(1)

    def gn_finalize(self, partial, Nimg, HW, Cc, groups, gamma, beta, eps, scale_out, shift_out,
                    mean_out=None, rstd_out=None, center_out=None):
        cpg = Cc // groups
        p = partial.double().sum(1).view(Nimg, groups, cpg, 2).sum(2)
        cnt = HW * cpg
        mean = p[..., 0] / cnt
        var = (p[..., 1] / cnt - mean * mean).clamp_min(0)
        rstd = (1.0 / torch.sqrt(var + eps)).float()
        sc = rstd.repeat_interleave(cpg, dim=1) * (gamma if gamma is not None else 1.0)
        scale_out.copy_(sc)
        if center_out is not None:
            center_out.copy_(mean.float().repeat_interleave(cpg, dim=1))
            shift_out.copy_((beta if beta is not None else torch.zeros(Cc)).expand(Nimg, Cc))
        else:
            shift_out.copy_((beta if beta is not None else 0.0)
                            - mean.float().repeat_interleave(cpg, dim=1) * sc)
        if mean_out is not None:
            mean_out.copy_(mean.float())
        if rstd_out is not None:
            rstd_out.copy_(rstd)

    # ---- pools
    def gn_finalize_tiles(self, partial, tile_rows, Nimg, HW, Cc, groups, gamma, beta, eps,
                          scale_out, shift_out, mean_out=None, rstd_out=None, center_out=None):
        assert HW % tile_rows == 0
        T = HW // tile_rows
        p = partial.double().view(Nimg, T, Cc, 2)
        sums = p[..., 0]
        sumsq = p[..., 1] + sums * sums / tile_rows
        cpg = Cc // groups
        cnt = HW * cpg
        s = sums.sum(1).view(Nimg, groups, cpg).sum(2)
        q = sumsq.sum(1).view(Nimg, groups, cpg).sum(2)
        mean = s / cnt
        var = (q / cnt - mean * mean).clamp_min(0)
        rstd = (1.0 / torch.sqrt(var + eps)).float()
        sc = (gamma if gamma is not None else torch.ones(Cc)) * rstd.repeat_interleave(cpg, dim=1)
        scale_out.copy_(sc)
        b = beta if beta is not None else torch.zeros(Cc)
        if center_out is not None:
            center_out.copy_(mean.float().repeat_interleave(cpg, dim=1))
            shift_out.copy_(b.expand(Nimg, Cc))
        else:
            shift_out.copy_(b - mean.float().repeat_interleave(cpg, dim=1) * sc)
        if mean_out is not None:
            mean_out.copy_(mean.float())
        if rstd_out is not None:
            rstd_out.copy_(rstd)

    def maxpool3x3s2(self, x, y, N, H, W, Cc, Ho, Wo, in_scale=None, in_shift=None, in_relu=0,
                     in_center=None):
        if in_scale is not None:
            x = (x - in_center if in_center is not None else x) * in_scale + in_shift
            if in_relu:
                x = torch.relu(x)
        y.copy_(F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1))

    def scale_shift_add_act(self, x1, s1, t1, x2, s2, t2, y, M, Cc, act, c1=None, c2=None):
        a = x1.reshape(M, Cc) - (c1 if c1 is not None else 0.0)
        b = x2.reshape(M, Cc) - (c2 if c2 is not None else 0.0)
        v = a * s1 + t1 + b * s2 + t2
        y.view(M, Cc).copy_(_act(v, act))

    # ---- cached-feature DAgger data path
    @staticmethod
    def _pad_rows(src, offsets, B, Tmax, D, fill, dst):
        out = dst.view(Tmax, B, D)
        out.fill_(fill)
        rows = src.reshape(-1, D)
        for b in range(B):
            o, e = int(offsets[b]), int(offsets[b + 1])
            out[: e - o, b] = rows[o:e].to(dst.dtype)

    def ragged_pad_rows(self, src, offsets, B, Tmax, D, fill, dst):
        self._pad_rows(src, offsets, B, Tmax, D, fill, dst)

    def ragged_pad_rows_i64(self, src, offsets, B, Tmax, D, fill, dst):
        self._pad_rows(src, offsets, B, Tmax, D, fill, dst)

    def dagger_targets(self, oracle, offsets, B, Tmax, coef, corrected, weights, masks):
        corrected.view(Tmax, B).zero_()
        weights.view(Tmax, B).zero_()
        masks.view(Tmax, B).fill_(1)
        masks.view(Tmax, B)[0] = 0
        for b in range(B):
            o, e = int(offsets[b]), int(offsets[b + 1])
            a = oracle[o:e]
            infl = torch.ones(e - o, dtype=torch.bool)
            infl[1:] = a[1:] != a[:-1]
            corrected.view(Tmax, B)[: e - o, b] = a
            weights.view(Tmax, B)[: e - o, b] = torch.where(infl, torch.tensor(float(coef)),
                                                           torch.tensor(1.0))

    def ppo_loss(self, values, returns, value_preds, logp, old_logp, adv, ent_pano, ent_offset,
                 ent_distance, radians, B, clip, value_coef, entropy_coef, pano_coef, offset_coef,
                 distance_coef, reg_coef, use_clipped, stats, grads):
        """contract of vlnce_ppo_loss: torch's own formulas and autograd (ddppo_alg.py:78-121)"""
        with torch.enable_grad():   # (called from inside an autograd.Function's forward)
            leaves = [t.detach().clone().requires_grad_(True)
                      for t in (values, logp, ent_pano, ent_offset, ent_distance)]
            v, lp, ep, eo, ed = leaves
            entropy_loss = (pano_coef * ep + offset_coef * eo + distance_coef * ed).mean() * entropy_coef
            ratio = torch.exp(lp - old_logp)
            action_loss = -torch.min(ratio * adv, torch.clamp(ratio, 1 - clip, 1 + clip) * adv).mean()
            if use_clipped:
                vpc = value_preds + (v - value_preds).clamp(-clip, clip)
                value_loss = 0.5 * torch.max((v - returns).pow(2), (vpc - returns).pow(2)).mean()
            else:
                value_loss = 0.5 * (returns - v).pow(2).mean()
            value_loss = value_loss * value_coef
            offset_loss = reg_coef * radians.abs().mean() if radians is not None else torch.zeros(())
            loss = value_loss + action_loss + offset_loss - entropy_loss
            gs = torch.autograd.grad(loss, leaves, allow_unused=True)
        stats.copy_(torch.stack([loss, value_loss, action_loss, entropy_loss, ep.mean(), eo.mean(),
                                 ed.mean(), offset_loss]).detach().float())
        for k, g in enumerate(gs):
            grads.view(5, B)[k].copy_(g if g is not None else torch.zeros(B))

    def ppo_returns(self, rewards, value_preds, masks, next_value, returns, T, N, gamma, tau,
                    use_gae):
        r, v, m, ret = (t.view(-1, N) for t in (rewards, value_preds, masks, returns))
        if use_gae:
            v[T] = next_value.view(N)
            gae = torch.zeros(N)
            for s in reversed(range(T)):
                delta = r[s] + gamma * v[s + 1] * m[s + 1] - v[s]
                gae = delta + gamma * tau * m[s + 1] * gae
                ret[s] = gae + v[s]
        else:
            ret[T] = next_value.view(N)
            for s in reversed(range(T)):
                ret[s] = ret[s + 1] * gamma * m[s + 1] + r[s]

    def space_to_depth2(self, x, y, N, H, W, Cc, pad_lo, pad_hi, scale=None, shift=None):
        v = x.reshape(N, H, W, Cc)
        if scale is not None:
            v = v * scale + shift
        v = v.view(N, H // 2, 2, W // 2, 2, Cc).permute(0, 1, 3, 2, 4, 5).reshape(
            N, H // 2, W // 2, 4 * Cc)
        y.zero_()
        y.view(N, H // 2 + pad_lo + pad_hi, W // 2 + pad_lo + pad_hi, 4 * Cc)[
            :, pad_lo:pad_lo + H // 2, pad_lo:pad_lo + W // 2] = v

    def avgpool2x2(self, x, y, N, H, W, Cc):
        y.copy_(F.avg_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))

    def group_norm_small(self, x, N, HW, Cc, groups, gamma, beta, eps, residual, act, y):
        v = F.group_norm(x.reshape(N, HW, Cc).permute(0, 2, 1), groups, gamma, beta, eps)
        v = v.permute(0, 2, 1).reshape(y.shape)
        if residual is not None:
            v = v + residual.reshape(y.shape)
        y.copy_(_act(v, act))

    def gru_rollout_supported(self, N, H):
        return 0 < N <= 16 and H in (64, 128, 256, 512)

    def gru_rollout_workspace_bytes(self, N, H):
        return 2 * N * 3 * H * 8

    def gru_rollout_fwd(self, gi, h0, mask, w_hh, b_hh, hp, out, gates, aux, workspace, T, N, H):
        h = h0
        for t in range(T):
            hp[t] = h * mask[t].view(N, 1).float()
            gh = hp[t] @ w_hh.t() + b_hh
            self.gru_gates_fwd(gi[t], gh, hp[t], None, out[t], gates[t], aux[t], N, H)
            h = out[t]

    def gru_rollout_bwd(self, dout, dh_final, gates, aux, hp, mask, w_hh_t, dgi, dgh, dh0,
                        workspace, T, N, H):
        carry = torch.zeros(N, H) if dh_final is None else dh_final.clone()
        acc = torch.empty(N, H)
        for t in range(T - 1, -1, -1):
            d = carry if dout is None else dout.view(T, N, H)[t] + carry
            self.gru_gates_bwd(d, gates[t], aux[t], hp[t], None, dgi[t], dgh[t], acc, N, H)
            carry = (acc + dgh[t] @ w_hh_t.t()) * mask[t].view(N, 1).float()
        dh0.copy_(carry)

    def rnn_step_supported(self, N, H, lstm):
        return False  # the fused steps are a launch-count optimisation of the same arithmetic

    # ---- observation ingest (contract of include/vlnce_hip.h, vlnce_frames_*)
    @staticmethod
    def _frames_f32(fr):
        """the descriptor's frames as fp32 [N*(F+extra), H, W, C] (crop, stack, masked extra frame)"""
        N, F_, Hs, Ws, Cc = fr["N"], fr["F"], fr["Hs"], fr["Ws"], fr["C"]
        y0, x0, H, W = fr["y0"], fr["x0"], fr["H"], fr["W"]
        v = fr["x"].reshape(N, F_, Hs, Ws, Cc)[:, :, y0:y0 + H, x0:x0 + W].float()
        if fr.get("x2") is not None:
            e = fr["x2"].reshape(N, 1, Hs, Ws, Cc)[:, :, y0:y0 + H, x0:x0 + W].float()
            if fr.get("mask2") is not None:
                e = e * fr["mask2"].reshape(N, 1, 1, 1, 1).float()
            v = torch.cat([v, e], dim=1)
        return v.reshape(-1, H, W, Cc)

    def frames_s2d(self, fr, y, pad_lo, pad_hi, scale=None, shift=None):
        v = self._frames_f32(fr)
        self.space_to_depth2(v, y, v.size(0), fr["H"], fr["W"], fr["C"], pad_lo, pad_hi, scale, shift)

    def stem7_fwd(self, fr, in_scale, in_shift, w_frag, y, scale=None, shift=None, act=0, bn=None,
                  w_format=None):
        """vlnce_stem7_fwd: conv 7x7 / stride 2 / pad 3 of (frame * in_scale + in_shift), the
        filters given as B fragments [Cout/32][11][3][64 lanes][8 bf16] (k' = kh*24 + kw*3 + c)."""
        v = self._frames_f32(fr)
        if in_scale is not None:
            v = v * in_scale + in_shift
        Cout = y.size(-1)
        if (w_format or getattr(w_frag, "_vlnce_fmt", 1)) == 2:   # fp16 planes {h * 2^11, (w - h) * 2^11, h}
            pf = w_frag.view(torch.float16).view(Cout // 32, 11, 3, 2, 32, 8).float()
            pl = pf[:, :, 2] + pf[:, :, 1] / 2048.0
        else:
            pl = w_frag.view(torch.bfloat16).view(Cout // 32, 11, 3, 2, 32, 8).float().sum(2)  # nb ks half l31 e
        w = pl.permute(0, 3, 1, 2, 4).reshape(Cout, 176)[:, :168].reshape(Cout, 7, 24)[:, :, :21]
        w = w.reshape(Cout, 7, 7, 3).permute(0, 3, 1, 2)
        raw = F.conv2d(v.permute(0, 3, 1, 2), w, stride=2, padding=3).permute(0, 2, 3, 1)
        if bn is not None:
            flat = raw.reshape(-1, Cout).double()
            bn[0, :, 0] += flat.sum(0)
            bn[0, :, 1] += (flat ** 2).sum(0)
        out = raw
        if scale is not None:
            out = out * scale
        if shift is not None:
            out = out + shift
        y.copy_(_act(out, act))

    def frames_avgpool2(self, fr, y):
        v = self._frames_f32(fr)
        y.copy_(F.avg_pool2d(v.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))

    def frames_f32(self, fr, y, scale=None, shift=None):
        v = self._frames_f32(fr)
        y.copy_(v * scale + shift if scale is not None else v)

    def frames_gather(self, srcs, elem_bytes, N, Hs, Ws, Cc, y0, x0, H, W, out):
        out.copy_(torch.stack([t[:, y0:y0 + H, x0:x0 + W] for t in srcs], dim=1))

    def embedding_bwd(self, tokens, grad_rows, grad_weight, padding_idx):
        t = tokens.reshape(-1)
        keep = t != (-1 if padding_idx is None else padding_idx)
        grad_weight.index_add_(0, t[keep], grad_rows.reshape(t.numel(), -1)[keep])

    def frames_resize_area(self, x, is_u8, NF, Hs, Ws, Cc, OH, OW, y0, x0, H, W, out):
        v = x.reshape(NF, Hs, Ws, Cc).permute(0, 3, 1, 2).float()
        r = F.interpolate(v, size=(OH, OW), mode="area").to(x.dtype).permute(0, 2, 3, 1)
        out.copy_(r[:, y0:y0 + H, x0:x0 + W].reshape(out.shape))

    def adaptive_avgpool(self, x, y, N, H, W, Cc, OH, OW, ldy):
        v = F.adaptive_avg_pool2d(x.permute(0, 3, 1, 2), (OH, OW)).permute(0, 2, 3, 1)
        _mat(y, N * OH * OW, Cc, ldy).copy_(v.reshape(-1, Cc))

    def mean_rows(self, x, y, B, P, Cc):
        y.copy_(x.reshape(B, P, Cc).mean(1))

    # ---- attention
    # ---- categorical action head
    def action_head_fwd(self, x, ldx, w, b, M, K, A, logits_out, nan_count=None):
        z = _mat(x, M, K, ldx) @ w.t()
        if b is not None:
            z = z + b
        n = z - z.logsumexp(dim=-1, keepdim=True)
        logits_out.copy_(n)
        if nan_count is not None:
            nan_count += int((n != n).any(dim=1).sum())

    def action_head_bwd(self, x, ldx, w, logits, dlogits, M, K, A, dx=None, dw=None, db=None):
        dz = dlogits - logits.exp() * dlogits.sum(dim=1, keepdim=True)
        if dx is not None:
            dx.copy_(dz @ w)
        if dw is not None:
            dw.copy_(dz.t() @ _mat(x, M, K, ldx))
        if db is not None:
            db.copy_(dz.sum(0))

    @staticmethod
    def _kv(t, B, P, D, ld):
        return t.as_strided((B, P, D), (P * ld, ld, 1))

    def attn_fwd(self, q, K, ldk, V, ldv, mask, mask_mode, scale, out, attn_out, B, P, Dk, Dv,
                 kv_index=None):
        U = B if kv_index is None else int(kv_index.max()) + 1
        k = self._kv(K, U, P, Dk, ldk)
        v = self._kv(V, U, P, Dv, ldv)
        if kv_index is not None:   # K / V / mask blocks shared by groups of queries
            k, v = k[kv_index], v[kv_index]
            mask = mask[kv_index] if mask is not None else None
        logits = torch.einsum("bd,bpd->bp", q, k)
        if mask is not None and mask_mode == 1:
            logits = logits - mask.float() * 1e8
        if mask is not None and mask_mode == 2:
            logits = logits * mask.float()
        a = torch.softmax(logits * scale, dim=1)
        if attn_out is not None:
            attn_out.copy_(a)
        out.copy_(torch.einsum("bp,bpd->bd", a, v))

    def attn_bwd(self, dout, q, K, ldk, V, ldv, mask, mask_mode, scale, attn, dq, dK, lddk, dV,
                 lddv, B, P, Dk, Dv, kv_index=None):
        U = B if kv_index is None else int(kv_index.max()) + 1
        k = self._kv(K, U, P, Dk, ldk)
        v = self._kv(V, U, P, Dv, ldv)
        if kv_index is not None:
            k, v = k[kv_index], v[kv_index]
            mask = mask[kv_index] if mask is not None else None
        da = torch.einsum("bd,bpd->bp", dout, v)
        dl = attn * (da - (attn * da).sum(1, keepdim=True)) * scale
        if mask is not None and mask_mode == 2:
            dl = dl * mask.float()
        if dV is not None:
            self._kv(dV, B, P, Dv, lddv).copy_(attn.unsqueeze(2) * dout.unsqueeze(1))
        if dK is not None:
            self._kv(dK, B, P, Dk, lddk).copy_(dl.unsqueeze(2) * q.unsqueeze(1))
        if dq is not None:
            dq.copy_(torch.einsum("bp,bpd->bd", dl, k))

    def segment_sum(self, x, index, B, U, row_elems, out):
        out.view(U, row_elems).zero_().index_add_(0, index, x.reshape(B, row_elems))

    def rowzero_mask(self, x, ld, rows, Cc, mask):
        mask.view(-1).copy_((_mat(x, rows, Cc, ld) == 0).all(1).to(torch.uint8))

    # ---- recurrent cells
    def gru_gates_fwd(self, gi, gh, h_prev, mask, h_out, gates_out, hn_out, B, H):
        hp = h_prev if mask is None else h_prev * mask.view(B, 1).float()
        ir, iz, inn = gi.view(B, 3 * H).split(H, 1)
        hr, hz, hn = gh.view(B, 3 * H).split(H, 1)
        r, z = torch.sigmoid(ir + hr), torch.sigmoid(iz + hz)
        n = torch.tanh(inn + r * hn)
        h_out.copy_((1 - z) * n + z * hp)
        if gates_out is not None:
            gates_out.copy_(torch.cat([r, z, n], 1))
        if hn_out is not None:
            hn_out.copy_(hn)

    def gru_gates_bwd(self, dh_out, gates, hn, h_prev, mask, dgi, dgh, dh_prev, B, H):
        mk = 1.0 if mask is None else mask.view(B, 1).float()
        hp = h_prev * mk
        r, z, n = gates.split(H, 1)
        dn = dh_out * (1 - z)
        dz = dh_out * (hp - n)
        dnp = dn * (1 - n * n)
        drp = dnp * hn * r * (1 - r)
        dzp = dz * z * (1 - z)
        dgi.copy_(torch.cat([drp, dzp, dnp], 1))
        dgh.copy_(torch.cat([drp, dzp, dnp * r], 1))
        dh_prev.copy_(dh_out * z * mk)

    def lstm_gates_fwd(self, gi, gh, c_prev, mask, h_out, c_out, gates_out, B, H):
        cp = c_prev if mask is None else c_prev * mask.view(B, 1).float()
        i, f, g, o = (gi + gh).view(B, 4 * H).split(H, 1)
        i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
        c = f * cp + i * g
        c_out.copy_(c)
        h_out.copy_(o * torch.tanh(c))
        if gates_out is not None:
            gates_out.copy_(torch.cat([i, f, g, o], 1))

    def lstm_gates_bwd(self, dh_out, dc_out, gates, c_prev, c_out, mask, dgates, dc_prev, B, H):
        mk = 1.0 if mask is None else mask.view(B, 1).float()
        cp = c_prev * mk
        i, f, g, o = gates.split(H, 1)
        tc = torch.tanh(c_out)
        dh = dh_out if dh_out is not None else torch.zeros_like(c_out)
        dc = (dc_out if dc_out is not None else 0) + dh * o * (1 - tc * tc)
        dgates.copy_(torch.cat([dc * g * i * (1 - i), dc * cp * f * (1 - f), dc * i * (1 - g * g),
                                dh * tc * o * (1 - o)], 1))
        dc_prev.copy_(dc * f * mk)

    # ---- backward of the visual trunks (contracts of csrc/bwd.hip, vlnce_conv2d_wgrad)
    def conv2d_wgrad(self, x, dy, dw, g, dy_pow2=None, accumulate=False):
        N, H, W, Cin, Cout = g["N"], g["H"], g["W"], g["Cin"], g["Cout"]
        xi = x.as_strided((N, H, W, Cin), (H * W * g["ldx"], W * g["ldx"], g["ldx"], 1))
        xi = xi.permute(0, 3, 1, 2).contiguous()
        gy = dy.reshape(N, g["Ho"], g["Wo"], Cout).permute(0, 3, 1, 2).contiguous()
        gw = torch.nn.grad.conv2d_weight(xi, (Cout, Cin, g["KH"], g["KW"]), gy,
                                         stride=g["stride"], padding=g["pad"])
        if accumulate:
            dw.view(Cout, g["KH"], g["KW"], Cin).add_(gw.permute(0, 2, 3, 1))
        else:
            dw.view(Cout, g["KH"], g["KW"], Cin).copy_(gw.permute(0, 2, 3, 1))

    def bn_bwd_workspace_floats(self, M, Cc):
        return 1

    def bn_bwd(self, dy, y, x, mean, rstd, gamma, M, Cc, relu, use_batch_stats, dx, dres, dgamma,
               dbeta, workspace=None, pow2=None):
        g = dy.reshape(M, Cc)
        if relu:
            g = g * (y.reshape(M, Cc) > 0)
        xh = (x.reshape(M, Cc) - mean) * rstd
        db, dg = g.sum(0), (g * xh).sum(0)
        dbeta.copy_(db)
        dgamma.copy_(dg)
        k = (gamma if gamma is not None else 1.0) * rstd
        v = g - db / M - xh * dg / M if use_batch_stats else g
        dx.view(M, Cc).copy_(k * v)
        self._fill_pow2(pow2, dx)
        if dres is not None:
            dres.view(M, Cc).copy_(g)

    def gn_bwd_workspace_floats(self, Nimg, HW, Cc, groups):
        return 1

    @staticmethod
    def _fill_pow2(pow2, dx):
        """the simulator takes the true maximum where the kernels take an upper bound"""
        if pow2 is None:
            return
        m = float(dx.abs().max())
        up = 1.0 if not (m > 0) else 2.0 ** (14 - math.frexp(m)[1])
        pow2[0].fill_(up)
        pow2[1].fill_(1.0 / up)

    def gn_bwd(self, dy, y, x, mean, rstd, gamma, Nimg, HW, Cc, groups, relu, dx, dres, dgamma,
               dbeta, workspace, pow2=None):
        cpg = Cc // groups
        g = dy.reshape(Nimg, HW, groups, cpg)
        if relu:
            g = g * (y.reshape(Nimg, HW, groups, cpg) > 0)
        xh = (x.reshape(Nimg, HW, groups, cpg) - mean.view(Nimg, 1, groups, 1)) * rstd.view(
            Nimg, 1, groups, 1)
        ga = (gamma if gamma is not None else torch.ones(Cc)).view(1, 1, groups, cpg)
        dbeta.copy_(g.sum((0, 1)).reshape(Cc))
        dgamma.copy_((g * xh).sum((0, 1)).reshape(Cc))
        cnt = HW * cpg
        s1 = (g * ga).sum((1, 3), keepdim=True)
        s2 = (g * ga * xh).sum((1, 3), keepdim=True)
        v = rstd.view(Nimg, 1, groups, 1) * (g * ga - s1 / cnt - xh * s2 / cnt)
        dx.view(Nimg, HW, groups, cpg).copy_(v)
        self._fill_pow2(pow2, dx)
        if dres is not None:
            dres.view(Nimg, HW, groups, cpg).copy_(g)

    def maxpool3x3s2_argmax(self, x, y, argmax, N, H, W, Cc, Ho, Wo):
        xp = F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1), value=float("-inf"))
        win = xp.unfold(2, 3, 2).unfold(3, 3, 2).reshape(N, Cc, Ho, Wo, 9)
        m, a = win.max(dim=4)
        # torch.max returns the first maximal index for ties on CPU, matching the scan order;
        # padded (-inf) taps can only win when the whole window is -inf, which cannot happen
        y.copy_(m.permute(0, 2, 3, 1))
        argmax.copy_(a.permute(0, 2, 3, 1).to(torch.uint8))

    def maxpool3x3s2_bwd(self, dy, argmax, dx, N, H, W, Cc, Ho, Wo):
        gx = torch.zeros(N, Cc, H + 2, W + 2)
        a = argmax.permute(0, 3, 1, 2).long()
        gy = dy.permute(0, 3, 1, 2)
        ho = torch.arange(Ho).view(1, 1, Ho, 1)
        wo = torch.arange(Wo).view(1, 1, 1, Wo)
        hi = (ho * 2 + a // 3).expand_as(a)
        wi = (wo * 2 + a % 3).expand_as(a)
        n = torch.arange(N).view(N, 1, 1, 1).expand_as(a)
        c = torch.arange(Cc).view(1, Cc, 1, 1).expand_as(a)
        gx.index_put_((n, c, hi, wi), gy, accumulate=True)
        dx.copy_(gx[:, :, 1:-1, 1:-1].permute(0, 2, 3, 1))

    def adaptive_avgpool_bwd(self, dy, dx, N, H, W, Cc, OH, OW):
        with torch.enable_grad():
            xin = torch.zeros(N, Cc, H, W, requires_grad=True)
            out = F.adaptive_avg_pool2d(xin, (OH, OW))
            (gx,) = torch.autograd.grad(out, xin, dy.permute(0, 3, 1, 2).contiguous())
        dx.copy_(gx.permute(0, 2, 3, 1))

    # ---- packed-sequence RNN (contract of csrc/rnn_seq.hip)
    def rnn_seq_supported(self, kind, H):
        return kind in (0, 1) and H in (64, 128)

    @staticmethod
    def _time_index(lengths, s, reverse):
        active = s < lengths
        tt = (lengths - 1 - s) if reverse else torch.full_like(lengths, s)
        return active, tt.clamp_min(0)

    def rnn_seq_fwd(self, kind, dirs, gi, w_hh, b_hh, lengths, out, h_final, gates_save, aux_save,
                    B, Lm, H):
        G = 4 if kind == 0 else 3
        ar = torch.arange(B)
        ln = lengths.long()
        for d in range(dirs):
            h = torch.zeros(B, H)
            c = torch.zeros(B, H)
            for s in range(Lm):
                active, tt = self._time_index(ln, s, d == 1)
                x = gi[d][tt, ar]  # [B, G*H]
                gh = h @ w_hh[d].t() + b_hh[d]
                if kind == 0:
                    i, f, g, o = (x + gh).split(H, 1)
                    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
                    cn = f * c + i * g
                    hn_ = o * torch.tanh(cn)
                    gs, ax = torch.cat([i, f, g, o], 1), cn
                else:
                    xr, xz, xn = x.split(H, 1)
                    hr, hz, hn = gh.split(H, 1)
                    r, z = torch.sigmoid(xr + hr), torch.sigmoid(xz + hz)
                    n = torch.tanh(xn + r * hn)
                    hn_ = (1 - z) * n + z * h
                    cn = c
                    gs, ax = torch.cat([r, z, n], 1), hn
                a = active.view(B, 1)
                idx = (tt[active], ar[active])
                out[d][idx] = hn_[active]
                if gates_save is not None:
                    gates_save[d][idx] = gs[active]
                    aux_save[d][idx] = ax[active]
                h = torch.where(a, hn_, h)
                c = torch.where(a, cn, c)
            h_final[d].copy_(h)

    def rnn_seq_bwd(self, kind, dirs, w_hh_t, lengths, out, gates_save, aux_save, dout, dh_final,
                    dgi, dgh, B, Lm, H):
        ar = torch.arange(B)
        ln = lengths.long()
        for d in range(dirs):
            rev = d == 1
            dh = dh_final[d].clone() if dh_final is not None and dh_final[d] is not None \
                else torch.zeros(B, H)
            dc = torch.zeros(B, H)
            W = w_hh_t[d].t()  # [G*H, H]
            for s in range(Lm - 1, -1, -1):
                active, tt = self._time_index(ln, s, rev)
                tp = (tt + 1) if rev else (tt - 1)
                tp = tp.clamp(0, Lm - 1)
                a = active.view(B, 1)
                gs = gates_save[d][tt, ar]
                ax = aux_save[d][tt, ar]
                dht = dh + (dout[d][tt, ar] if dout is not None and dout[d] is not None else 0)
                if kind == 0:
                    i, f, g, o = gs.split(H, 1)
                    cp = aux_save[d][tp, ar] if s > 0 else torch.zeros(B, H)
                    tc = torch.tanh(ax)
                    dct = dc + dht * o * (1 - tc * tc)
                    dpre = torch.cat([dct * g * i * (1 - i), dct * cp * f * (1 - f),
                                      dct * i * (1 - g * g), dht * tc * o * (1 - o)], 1)
                    dpre_h = dpre
                    dc = torch.where(a, dct * f, dc)
                    keep = 0
                else:
                    r, z, n = gs.split(H, 1)
                    hp = out[d][tp, ar] if s > 0 else torch.zeros(B, H)
                    dn = dht * (1 - z)
                    dz = dht * (hp - n)
                    dnp = dn * (1 - n * n)
                    drp = dnp * ax * r * (1 - r)
                    dzp = dz * z * (1 - z)
                    dpre = torch.cat([drp, dzp, dnp], 1)
                    dpre_h = torch.cat([drp, dzp, dnp * r], 1)
                    keep = dht * z
                idx = (tt[active], ar[active])
                dgi[d][idx] = dpre[active]
                if kind == 1:
                    dgh[d][idx] = dpre_h[active]
                dh = torch.where(a, dpre_h @ W + keep, dh)

    # ---- skinny linear layers (contract of csrc/linear_rows.hip)
    def linear_rows_supported(self, M, N, K):
        return 1 <= M <= 128 and N % 4 == 0 and K % 4 == 0 and K >= 4

    @staticmethod
    def _act_grad(g, y, act):
        if act == 1:
            return g * (y > 0)
        if act == 2:
            return g * y * (1 - y)
        if act == 3:
            return g * (1 - y * y)
        return g

    def linear_rows_fwd(self, x, ldx, w, ldw, bias, act, y, ldy, M, N, K):
        xv = torch.as_strided(x, (M, K), (ldx, 1), x.storage_offset())
        wv = torch.as_strided(w, (N, K), (ldw, 1), w.storage_offset())
        v = xv @ wv.t()
        if bias is not None:
            v = v + bias
        v = {0: lambda t: t, 1: torch.relu, 2: torch.sigmoid, 3: torch.tanh}[act](v)
        torch.as_strided(y, (M, N), (ldy, 1), y.storage_offset()).copy_(v)

    def linear_rows_bwd(self, x, ldx, w, ldw, dy, lddy, y, ldy, act, dx, dw, db, M, N, K):
        xv = torch.as_strided(x, (M, K), (ldx, 1), x.storage_offset())
        wv = torch.as_strided(w, (N, K), (ldw, 1), w.storage_offset())
        dz = torch.as_strided(dy, (M, N), (lddy, 1), dy.storage_offset())
        if act != 0:
            dz = self._act_grad(dz, torch.as_strided(y, (M, N), (ldy, 1), y.storage_offset()), act)
        if dx is not None:
            dx.copy_(dz @ wv)
        if dw is not None:
            dw.copy_(dz.t() @ xv)
        if db is not None:
            db.copy_(dz.sum(0))

    # second-generation entry points (ABI 141): same contracts, self-zeroing, consumer-layout outputs
    def rnn_seq_fwd2(self, kind, dirs, gi, w_hh, b_hh, lengths, out_tm, seq, seq_st, seq_sb, h_final,
                     gates_save, aux_save, B, Lm, H):
        for o in out_tm:
            o.zero_()
        self.rnn_seq_fwd(kind, dirs, gi, w_hh, b_hh, lengths, out_tm, h_final, gates_save, aux_save,
                         B, Lm, H)
        if seq is not None:
            for d in range(dirs):
                view = torch.as_strided(seq, (Lm, B, H), (seq_st, seq_sb, 1), seq.storage_offset() + d * H)
                view.copy_(out_tm[d])

    def rnn_seq_bwd2(self, kind, dirs, w_hh, lengths, out_tm, gates_save, aux_save, dseq, dseq_st,
                     dseq_sb, dout_ws, dh_final, dgi, dgh, B, Lm, H):
        dout = None
        if dseq is not None:
            dout = [torch.as_strided(dseq, (Lm, B, H), (dseq_st, dseq_sb, 1), dseq.storage_offset() + d * H)
                    for d in range(dirs)]
        for t in list(dgi) + (list(dgh) if dgh is not None else []):
            t.zero_()
        self.rnn_seq_bwd(kind, dirs, [w.t() for w in w_hh], lengths, out_tm, gates_save, aux_save,
                         dout, dh_final, dgi, dgh, B, Lm, H)

    def rnn_seq_wgrad(self, kind, dirs, dgi, dgh, out_tm, x_tm, ldx, E, w_ih, dw_ih, dw_hh, db_ih,
                      db_hh, dx_tm, B, Lm, H, first_dir=0):
        GH = dgi[0].shape[-1]
        x = x_tm.reshape(Lm * B, -1)[:, :E]
        if dx_tm is not None:
            dx_tm.zero_()
        for d in range(dirs):
            dG = (dgh[d] if kind == 1 else dgi[d]).reshape(Lm, B, GH)
            hprev = torch.zeros(Lm, B, H)
            if Lm > 1:
                if first_dir + d == 1:
                    hprev[:-1] = out_tm[d][1:]
                else:
                    hprev[1:] = out_tm[d][:-1]
            dw_hh[d].copy_(dG.reshape(-1, GH).t() @ hprev.reshape(-1, H))
            db_hh[d].copy_(dG.reshape(-1, GH).sum(0))
            g = dgi[d].reshape(-1, GH)
            dw_ih[d].copy_(g.t() @ x)
            db_ih[d].copy_(g.sum(0))
            if dx_tm is not None:
                dx_tm += g @ w_ih[d]

    def mask_rows(self, x, mask, out, B, H):
        out.copy_(x.view(B, H) * mask.view(B, 1).float())

    def select_rows(self, mask, a, b, out, B, H):
        z = torch.zeros(B, H)
        out.copy_(torch.where(mask.view(B, 1) != 0, a if a is not None else z,
                              b if b is not None else z))

    def act_bwd(self, dy, y, dz, n, act):
        if act == 1:
            r = dy * (y > 0)
        elif act == 2:
            r = dy * y * (1 - y)
        elif act == 3:
            r = dy * (1 - y * y)
        else:
            r = dy
        dz.copy_(r)
