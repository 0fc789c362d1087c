"""PEFT-method halves of the launch-plan engine: AdaptFormer adapters, MeLO / LoRA, unfrozen-backbone gradients (`fft` / `bitfit`), EVP, DVPT and SSF
(adaptformer.py, melo.py, train.py:123-137, evp.py, dvpt.py, ssf.py).  Mixed into engine.Engine; every method only enqueues C-ABI launches."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch

from . import lib as L
from . import ops
from .engine_common import (SEED_EMB, SEED_LAYER, SEED_PROMPT, GRAPH_WARMUP, Names, PLAN_TIMING, SIDE_STREAM_PRIORITY, STEP_MODE, USE_GRAPHS, _ABLATE, _EPI_NAMES, _FIX_IN_LN, _LOC_SHIFT, _MODE, _SIDE_STREAMS, _on, evp_highpass_operator)  # noqa: F401


class PeftPaths:
    # ---- AdaptFormer (adaptformer.py:58-78, 93-97): r = up(ReLU(down(LN_a(x)))), x_out = ff(x) + x + r -----------------------
    def _adapter_prefix(self, i):
        return f"transformer.layers.{i}.1"

    def _adapter_shadows(self, train):
        """bf16 MFMA operands of the TRAINABLE adapter weights, refreshed in place every step (inside the captured graph)."""
        w = self._w16
        for i in range(self.depth):
            p = self._adapter_prefix(i)
            wd, wu = self._d(p + ".down_adapter_proj.weight"), self._d(p + ".up_adapter_proj.weight")
            w[f"ad_d{i}"] = ops.to_operand(wd, None if self.fp32 else w.get(f"ad_d{i}"), self.adt)
            w[f"ad_u{i}"] = ops.to_operand(wu, None if self.fp32 else w.get(f"ad_u{i}"), self.adt)
            if train:
                w[f"ad_dT{i}"] = ops.transpose_operand(wd, w.get(f"ad_dT{i}"), self.adt)
                w[f"ad_uT{i}"] = ops.transpose_operand(wu, w.get(f"ad_uT{i}"), self.adt)

    def _adapter_fwd_down(self, ws, i, si, g1, M):
        p, d, ad = self._adapter_prefix(i), self._d, ws["ad"][si]
        ops.layernorm_fwd(g1, d(p + ".adapter_layer_norm_before.weight"), d(p + ".adapter_layer_norm_before.bias"), M, self.C, y16=ws["xa"],
                          mean=ad["mean"], rstd=ad["rstd"])
        self._gemm(ws["xa"], self._w16[f"ad_d{i}"], M, ad["h16"], epilogue=ops.EPI_BIAS_RELU_BF16, bias=d(p + ".down_adapter_proj.bias"))

    def _adapter_fwd_up(self, ws, i, si, gout, M):
        p = self._adapter_prefix(i)
        self._gemm(ws["ad"][si]["h16"], self._w16[f"ad_u{i}"], M, gout, epilogue=ops.EPI_BIAS_RES_F32, bias=self._d(p + ".up_adapter_proj.bias"),
                   res=gout)

    def _adapter_bwd(self, ws, gv, i, dGout, dG1, M, refresh_operand=False):
        p, d, C, A = self._adapter_prefix(i), self._d, self.C, self.adim
        ad, w, sc = ws["ad"][i], self._w16, ws["scratch"]
        g, b = d(p + ".adapter_layer_norm_before.weight"), d(p + ".adapter_layer_norm_before.bias")
        if refresh_operand:
            # backbone dropout live (freeze_vit=False): the operand buffer holds dGout * mask(fc2's dropout) for the FFN branch; the adapter
            # branch joins the stream unmasked (adaptformer.py:96-98: x = ff(x) + x + adapter(x)), so it needs the plain gradient again
            ops.to_operand(dGout, ws["dG16"], self.adt)
        # up-projection: dh = (dGout . Wu) * [h > 0]; dWu = dGout^T . h; dbu = colsum(dGout)
        self._gemm(ws["dG16"], w[f"ad_uT{i}"], M, ws["dh16"], epilogue=ops.EPI_RELU_BWD_BF16, aux=ad["h16"])
        ops.cast_bf16_f32_strided(ad["h16"], ws["h32"], M, A, A)
        ops.cast_bf16_f32_strided(ws["dh16"], ws["dh32"], M, A, A)
        ops.outer_reduce(narrow=ws["h32"], wide=dGout, scratch=sc, out=gv[p + ".up_adapter_proj.weight"], colsum=gv[p + ".up_adapter_proj.bias"],
                         M=M, C=C, L=A, transposed=1, accumulate=0)
        # down-projection: dxa = dh . Wd; dWd = dh^T . LN_a(G1); dbd = colsum(dh)
        self._gemm(ws["dh16"], w[f"ad_dT{i}"], M, ws["dx32"], epilogue=ops.EPI_STORE_F32)
        ops.outer_reduce(narrow=ws["dh32"], wide=ws["G1"][i], mean=ad["mean"], rstd=ad["rstd"], ln_gamma=g, ln_beta=b, scratch=sc,
                         out=gv[p + ".down_adapter_proj.weight"], M=M, C=C, L=A, transposed=0, accumulate=0)
        ops.colsum(ws["dh32"], gv[p + ".down_adapter_proj.bias"], sc, M, A)
        # trainable LayerNorm in front of the adapter: input gradient accumulates into dG1, affine gradients
        ops.layernorm_bwd(ws["dx32"], ws["G1"][i], ad["mean"], ad["rstd"], g, M, C, dx=dG1, dres=dG1, dx16=ws["dG16"])
        ops.layernorm_bwd_affine(ws["dx32"], ws["G1"][i], ad["mean"], ad["rstd"], gv[p + ".adapter_layer_norm_before.weight"],
                                 gv[p + ".adapter_layer_norm_before.bias"], sc, M, C)

    # ---- MeLO / LoRA (melo.py:41-47): qkv = W x + s B_q A_q x (q columns) + s B_v A_v x (v columns) -------------------------
    def _lora_names(self, i):
        q = self.names.attn(i) + ".to_qkv"
        return q + ".linear_a_q.weight", q + ".linear_b_q.weight", q + ".linear_a_v.weight", q + ".linear_b_v.weight"

    def _melo_merge(self, ws, train):
        """Fold the rank-r update into the bf16 QKV operand (and its transpose) every step: the forward is then the plain GEMM."""
        w, C = self._w16, self.C
        for i in range(self.depth):
            if i not in self.lora_layers:                   # a layer outside lora_layer (melo.py:67-68) keeps the plain frozen shadows
                continue                                    # refresh_weights() built for it
            aq, bq, av, bv = (self._d(n) for n in self._lora_names(i))
            ops.lora_merge(self._d(self.names.qkv_weight(i)), aq, bq, av, bv, ws["merge32"], C, self.r, self.lora_s)
            if self.fp32 and not isinstance(w.get(f"qkv{i}_own"), torch.Tensor):
                w[f"qkv{i}_own"] = torch.empty_like(ws["merge32"])     # the merged weight needs its own buffer per layer
            w[f"qkv{i}"] = ops.to_operand(ws["merge32"], w[f"qkv{i}_own"] if self.fp32 else w.get(f"qkv{i}"), self.adt)
            if train:
                w[f"qkv{i}_t"] = ops.transpose_operand(ws["merge32"], w.get(f"qkv{i}_t"), self.adt)

    def _melo_bwd(self, ws, gv, i, M):
        """dB = s dq^T u, dA = s (dq B)^T LN(x), u = LN(x) A^T -- all rank-r fp32 kernels over the bf16 dq / dv blocks."""
        C, r, d, sc = self.C, self.r, self._d, ws["scratch"]
        na_q, nb_q, na_v, nb_v = self._lora_names(i)
        a = self.names.attn(i)
        g1, b1, st, x = d(a + ".norm.weight"), d(a + ".norm.bias"), ws["stat"][i], ws["G"][i]
        lu = ws["lu"]
        ops.cast_bf16_f32_strided(ws["dqkv"], ws["dq32"], M, C, 3 * C, col0=0)
        ops.cast_bf16_f32_strided(ws["dqkv"], ws["dv32"], M, C, 3 * C, col0=2 * C)
        for na, nb, dblk, u, du in ((na_q, nb_q, ws["dq32"], lu["uq"], lu["duq"]), (na_v, nb_v, ws["dv32"], lu["uv"], lu["duv"])):
            ops.skinny_down(x=x, w=d(na), ln_gamma=g1, ln_beta=b1, y=u, M=M, C=C, L=r, act=0, w_layout=0, eps=1e-5)
            ops.outer_reduce(narrow=u, wide=dblk, scratch=sc, out=gv[nb], M=M, C=C, L=r, transposed=1, accumulate=0)
            ops.skinny_down(x=dblk, w=d(nb), y=du, M=M, C=C, L=r, act=0, w_layout=1)
            ops.outer_reduce(narrow=du, wide=x, mean=st[0], rstd=st[1], ln_gamma=g1, ln_beta=b1, scratch=sc, out=gv[na], M=M, C=C, L=r,
                             transposed=0, accumulate=0)
            if self.lora_s != 1:
                ops.scale_(gv[na], float(self.lora_s))
                ops.scale_(gv[nb], float(self.lora_s))

    # ---- unfrozen ViT tensors (`bitfit` / `fft`, train.py:123-137): biases and LayerNorm affines from column sums, weights from
    #      wgrad GEMMs dW = dY^T . X run as NT GEMMs over the transposed operands (contraction over the padded token count) -------
    def _bb_buffers(self, ws, B, device, wgrad):
        if "bbw" not in ws:
            C, M = self.C, B * self.T
            n = max(self.mlp, 3 * C, self.Kp)
            ws["bbw"] = dict(ones=torch.ones(n, device=device), zeros=torch.zeros(n, device=device), junk=torch.zeros(2 * n, device=device),
                             scratch=torch.zeros(64 * 2 * n, device=device), stat=[torch.zeros(M, device=device), torch.zeros(M, device=device)])
            ws["dyd"] = ops.act_zeros(M, C, torch.float32, device)           # dropout-masked copy of a layer gradient (bias / weight-gradient operand)
            ws["pg32"] = ops.act_zeros(B * self.N, C, torch.float32, device)  # patch rows of the input gradient (+ GAViKO's local-stream share)
        if wgrad and "sav" not in ws:
            C, M, Mp = self.C, B * self.T, ops.pad_rows(B * self.T)
            z = lambda r, c: ops.act_zeros(r, c, self.adt, device)
            ws["sav"] = dict(xn1=[z(M, C) for _ in range(self.depth)], xn2=[z(M, C) for _ in range(self.depth)],
                             act=[z(M, self.ldx) for _ in range(self.depth)])     # (row stride of the hidden buffers: mlp, + 64 with the GPA columns)
            ws["tA"] = z(max(self.mlp, 3 * C), Mp)
            ws["tB"] = z(max(self.ldx, self.Kp), Mp)
            ws["pg16"] = z(B * self.N, C)

    def _bb_wgrad(self, ws, dy_op, x_op, out, M, N, K, ldx=None):
        """out [N][K] (fp32) = dy_op[0:M, 0:N]^T . x_op[0:M, 0:K]; rows >= M of both operand buffers are zero by construction.
        ldx: row stride of x_op when it carries more than K columns (the MLP hidden buffer with the GPA columns behind it)."""
        Mp = ops.pad_rows(M)
        Kx = K if ldx is None else ldx
        tA, tB = ws["tA"].view(-1)[: ops.pad_rows(N) * Mp].view(-1, Mp), ws["tB"].view(-1)[: ops.pad_rows(Kx) * Mp].view(-1, Mp)
        if self.kind == "vpt" and self.deep and Mp > M:
            # deep VPT's sequence shrinks layer by layer (vpt.py:147-153) while the operand buffers are shared: rows M.. of this layer's
            # operands still hold a longer layer's values, and the contraction runs over the padded row count
            ops.memset_zero(dy_op.view(-1)[M * N: Mp * N])
            ops.memset_zero(x_op.view(-1)[M * Kx: Mp * Kx])
        ops.transpose_any(dy_op, tA, Mp, N)
        ops.transpose_any(x_op, tB, Mp, Kx)
        self._gemm(tA, tB[:K], N, out.view(N, K), epilogue=ops.EPI_STORE_F32)

    def _bb_linear_grads(self, ws, gv, bb, prefix, dy32, dy_op, x_op, M, N, K, ldx=None):
        """db = colsum(dy), dW = dy^T . x for one Linear of the backbone, for whichever of the two trains."""
        bw = ws["bbw"]
        if prefix + ".bias" in bb:
            src = dy32 if dy32 is not None else dy_op
            ops.colsum_any(src, gv[prefix + ".bias"], bw["ones"][:N], bw["zeros"][:N], bw["junk"][:N], bw["scratch"], M, N)
        if prefix + ".weight" in bb:
            self._bb_wgrad(ws, dy_op, x_op, gv[prefix + ".weight"], M, N, K, ldx=ldx)

    def _bb_ln_grads(self, ws, gv, bb, prefix, dy, x, mean, rstd, M):
        wn, bn = prefix + ".weight", prefix + ".bias"
        if wn in bb or bn in bb:
            C, junk = self.C, ws["bbw"]["junk"]
            ops.layernorm_bwd_affine(dy, x, mean, rstd, gv[wn] if wn in bb else junk[:C], gv[bn] if bn in bb else junk[C: 2 * C], ws["scratch"], M, C)

    def _bb_embed_grads(self, ws, gv, bb, dG0, B, dlocal=None, dlocal_to_pos=True):
        """pos_embedding / cls_token (batch sums of the input gradient), conv_proj bias and weight (the patch rows).  Rows: [cls | patches]
        for the plain layout, [P prompts | cls | patches] for GAViKO (gaviko.py:536-548), whose local stream = conv(img) + pos[1:]
        (gaviko.py:545-546) hands the patch rows a second gradient, `dlocal` [B*N][C] (the MWSA chain's input gradient).  EVP's `dlocal` is
        the embedding_generator's share of the RAW conv output (evp.py:347-348): it reaches the conv tensors but not pos_embedding."""
        C, T, N, bw = self.C, self.T, self.N, ws["bbw"]
        nm = self.names
        # row of the cls token (VPT: [cls | prompts | patches], vpt.py:127-131); the patch rows start at self.row_off
        r_cls = 0 if self.kind == "vpt" else self.row_off - 1
        pe, ct = nm.root + "pos_embedding", nm.root + "cls_token"
        cw, cb = nm.conv() + ".weight", nm.conv() + ".bias"
        need_rows = dlocal is not None and (pe in bb or cb in bb or cw in bb)
        if need_rows or cw in bb:
            ops.rows_gather(dG0, ws["pg32"], B, T, N, C, self.row_off)          # the patch rows of the input gradient, [B*N][C]
            if dlocal is not None:
                ops.add2d(ws["pg32"], C, dlocal, C, ws["pg32"], C, B * N, C)
        if pe in bb:
            pos = gv[pe].view(N + 1, C)
            ops.rows_batch_sum(dG0, pos[:1], None, B, T, r_cls, 1, C)
            if dlocal is None or not dlocal_to_pos:
                ops.rows_batch_sum(dG0, pos[1:], None, B, T, self.row_off, N, C)
            else:
                ops.rows_batch_sum(ws["pg32"], pos[1:], None, B, N, 0, N, C)
        if ct in bb:
            ops.rows_batch_sum(dG0, gv[ct].view(1, C), None, B, T, r_cls, 1, C)
        if cb in bb:
            if dlocal is None:
                ops.colsum_any(dG0, gv[cb], bw["ones"][:C], bw["zeros"][:C], bw["junk"][:C], bw["scratch"], B * N, C, rows_in=N, rows_out=T,
                               row_off=self.row_off)
            else:
                ops.colsum_any(ws["pg32"], gv[cb], bw["ones"][:C], bw["zeros"][:C], bw["junk"][:C], bw["scratch"], B * N, C)
        if cw in bb:
            ops.to_operand(ws["pg32"], ws["pg16"], self.adt)
            self._bb_wgrad(ws, ws["pg16"], ws["cols"], gv[cw].view(C, self.Kp), B * N, C, self.Kp)

    # ---- EVP (evp.py): prompts from a high-pass copy of the volume + the patch embeddings, added in front of every layer --------
    def _evp_state(self, device):
        """Static per-engine buffers: the high-pass operator, zero-padded copies of the trainable prompt-generator weights and of
        their gradients (the latents have rank r = dim/32 = 6 / 24 / 32; the rank-L kernels run at the padded width Lp)."""
        st = self.__dict__.get("_evp")
        if st is not None:
            return st
        C, Lp, Kp = self.C, self.Lp, self.Kp
        mk = lambda *s_: torch.zeros(s_, device=device)
        D, H, W = (g * p_ for g, p_ in zip(self.grid, self.patch))
        hp, dm = evp_highpass_operator(D, H, W, self.freq)
        st = dict(hp=torch.from_numpy(hp).to(device), dmask=torch.from_numpy(dm).to(device),
                  Wp=mk(64, Kp), bp=mk(64), We=mk(Lp, C), be=mk(Lp), Ws=mk(C, Lp),
                  Wi=[mk(Lp, Lp) for _ in range(self.depth)], WiT=[mk(Lp, Lp) for _ in range(self.depth)], bi=[mk(Lp) for _ in range(self.depth)],
                  dWs=mk(C, Lp), dWe=mk(Lp, C), dWp=mk(Lp, Kp), dvec=mk(Lp), dWi=mk(Lp, Lp))
        self.__dict__["_evp"] = st
        return st

    def _evp_latents(self, ws, B):
        """s = proj(highpass(img)) + embedding_generator(conv(img))  (evp.py:76-84, 347-349), at the padded width."""
        st, ev, d = self._evp_state(ws["img"].device), ws["ev"], self._d
        r, Lp, C, Kp, BN = self.r, self.Lp, self.C, self.Kp, B * self.N
        pg = "prompt_generator."
        ops.pad2d(d(pg + "prompt_generator.proj.weight").reshape(r, Kp), r, Kp, st["Wp"], 64, Kp)
        ops.pad2d(d(pg + "prompt_generator.proj.bias"), 1, r, st["bp"], 1, 64)
        ops.pad2d(d(pg + "embedding_generator.weight"), r, C, st["We"], Lp, C)
        ops.pad2d(d(pg + "embedding_generator.bias"), 1, r, st["be"], 1, Lp)
        ops.pad2d(d(pg + "shared_mlp.weight"), C, r, st["Ws"], C, Lp)
        for i in range(self.depth):
            wi = d(pg + f"lightweight_mlp_{i}.0.weight")
            ops.pad2d(wi, r, r, st["Wi"][i], Lp, Lp)
            ops.pad2d(wi, r, r, st["WiT"][i], Lp, Lp, transpose=True)
            ops.pad2d(d(pg + f"lightweight_mlp_{i}.0.bias"), 1, r, st["bi"][i], 1, Lp)
        ops.skinny_down(x=ws["xc"], w=st["We"], bias=st["be"], y=ev["e"], M=BN, C=C, L=Lp, act=0, w_layout=0)
        ops.evp_highpass(ws["img"], st["hp"], st["dmask"], ws["hp"])
        ops.patchify(ws["hp"], ws["hcols"], self.patch)
        ops.gemm_nt(ws["hcols"], st["Wp"], BN, ws["hc"], epilogue=ops.EPI_STORE_F32, bias=st["bp"])      # fp32 GEMM in both precisions (1.6 GF)
        ops.add2d(ws["hc"], 64, ev["e"], Lp, ev["s"], Lp, BN, Lp)

    def _evp_add_prompt(self, ws, i, si, g, B):
        """prompt_i = shared_mlp(GELU(lightweight_mlp_i(s)))  (evp.py:86-95), added to the patch rows of the layer input."""
        st, ev, d = self._evp_state(g.device), ws["ev"], self._d
        Lp, C, BN = self.Lp, self.C, B * self.N
        ops.small_linear_fwd(ev["s"], st["Wi"][i], st["bi"][i], ev["pre"][si], BN, Lp, Lp)
        ops.gelu_fwd(ev["pre"][si], ev["u"][si])
        ops.skinny_up(lat=ev["u"][si], w=st["Ws"], bias=d("prompt_generator.shared_mlp.bias"), out=ev["tmp"], M=BN, C=C, L=Lp, w_layout=0)
        ops.rows_patch(g, ev["tmp"], None, B, self.T, self.N, C, 1, True)

    def _evp_bwd_layer(self, ws, gv, i, dG, B):
        """d prompt_i = the patch rows of the gradient of layer i's input; accumulates d shared_mlp over the layers and d s."""
        st, ev, bw = self._evp_state(dG.device), ws["ev"], ws["evb"]
        Lp, C, BN, r = self.Lp, self.C, B * self.N, self.r
        pg = "prompt_generator."
        top = i == self.depth - 1
        ops.rows_gather(dG, ev["tmp"], B, self.T, self.N, C, 1)
        ops.outer_reduce(narrow=ev["u"][i], wide=ev["tmp"], scratch=ws["scratch"], out=st["dWs"], colsum=gv[pg + "shared_mlp.bias"], M=BN, C=C, L=Lp,
                         transposed=1, accumulate=0 if top else 1)
        ops.skinny_down(x=ev["tmp"], w=st["Ws"], y=bw["du"], M=BN, C=C, L=Lp, act=0, w_layout=1)
        ops.gelu_bwd(bw["du"], ev["pre"][i], bw["dpre"])
        ops.reduce_batch([(bw["dpre"], ev["s"], st["dWi"], 0), (bw["dpre"], None, st["dvec"], 0)], ws["rscratch"])
        ops.pad2d(st["dWi"], r, r, gv[pg + f"lightweight_mlp_{i}.0.weight"], r, r, ld_src=Lp)
        ops.pad2d(st["dvec"], 1, r, gv[pg + f"lightweight_mlp_{i}.0.bias"], 1, r, ld_src=Lp)
        ops.small_linear_fwd(bw["dpre"], st["WiT"][i], None, bw["ds"] if top else bw["ds_tmp"], BN, Lp, Lp)       # d s = dpre . W_i
        if not top:
            ops.add2d(bw["ds"], Lp, bw["ds_tmp"], Lp, bw["ds"], Lp, BN, Lp)

    def _evp_bwd_finish(self, ws, gv, B):
        st, ev, bw = self._evp_state(ws["img"].device), ws["ev"], ws["evb"]
        Lp, C, BN, r, Kp = self.Lp, self.C, B * self.N, self.r, self.Kp
        pg = "prompt_generator."
        ops.pad2d(st["dWs"], C, r, gv[pg + "shared_mlp.weight"], C, r, ld_src=Lp)
        ops.outer_reduce(narrow=bw["ds"], wide=ws["xc"], scratch=ws["scratch"], out=st["dWe"], M=BN, C=C, L=Lp, transposed=0, accumulate=0)
        ops.pad2d(st["dWe"], r, C, gv[pg + "embedding_generator.weight"], r, C)
        ops.reduce_batch([(bw["ds"], None, st["dvec"], 0)], ws["rscratch"])
        ops.pad2d(st["dvec"], 1, r, gv[pg + "embedding_generator.bias"], 1, r, ld_src=Lp)
        ops.pad2d(st["dvec"], 1, r, gv[pg + "prompt_generator.proj.bias"], 1, r, ld_src=Lp)
        ops.outer_reduce(narrow=bw["ds"], wide=ws["hcols"], scratch=ws["scratch"], out=st["dWp"], M=BN, C=Kp, L=Lp, transposed=0, accumulate=0)
        ops.pad2d(st["dWp"], r, Kp, gv[pg + "prompt_generator.proj.weight"].view(r, Kp), r, Kp)

    # ---- DVPT (dvpt.py:24-63): share_MLP beside the MLP block ------------------------------------------------------------------
    def _dvpt_names(self, i):
        p = f"transformer.layers.{i}.0.prompt_proj"
        return p + ".prompt_key_proj_d", p + ".prompt_key_proj_u", p + ".prompt_gate"

    def _dvpt_fwd_latents(self, ws, i, si, g1, M, B):
        pd, pu, pg = self._dvpt_names(i)
        d, v = self._d, ws["dv"][si]
        ops.skinny_down(x=g1, w=d(pd + ".weight"), bias=d(pd + ".bias"), y=v["z"], M=M, C=self.C, L=self.Lat, act=0, w_layout=0, act_in=1)
        ops.dvpt_fwd(z=v["z"], enh=v["enh"], lse=v["lse"], B=B, T=self.T, P=self.P, L=self.Lat, C=self.C, scale=self.C ** -0.5)

    def _dvpt_fwd_up(self, ws, i, si, gout, M):
        pd, pu, pg = self._dvpt_names(i)
        d, v = self._d, ws["dv"][si]
        ops.skinny_up(lat=v["z"], lat_override=v["enh"], w=d(pu + ".weight"), bias=d(pu + ".bias"), alpha_ptr=d(pg), out=gout, M=M, C=self.C,
                      L=self.Lat, T=self.T, P=self.P, w_layout=0, accumulate=1)

    def _dvpt_bwd_latents(self, ws, gv, i, dGout, M, B):
        """dcomb = dGout . W_u;  dW_u, db_u (gate applied afterwards), dgate;  latent backward -> dz;  db_d, dW_d."""
        pd, pu, pg = self._dvpt_names(i)
        d, v, bw, C, Lt = self._d, ws["dv"][i], ws["dvb"], self.C, self.Lat
        ops.skinny_down(x=dGout, w=d(pu + ".weight"), y=bw["dcomb"], M=M, C=C, L=Lt, act=0, w_layout=1)
        ops.outer_reduce(narrow=v["z"], lat_override=v["enh"], wide=dGout, scratch=ws["scratch"], out=gv[pu + ".weight"], colsum=gv[pu + ".bias"],
                         M=M, C=C, L=Lt, T=self.T, P=self.P, transposed=1, accumulate=0)
        ops.dvpt_bwd(z=v["z"], enh=v["enh"], lse=v["lse"], dcomb=bw["dcomb"], gate=d(pg), bu=d(pu + ".bias"), colsum_dy=gv[pu + ".bias"],
                     delta=bw["delta"], dz=bw["dz"], dgate=gv[pg], B=B, T=self.T, P=self.P, L=Lt, C=C, scale=C ** -0.5)
        ops.scale_dev_(gv[pu + ".weight"], d(pg))
        ops.scale_dev_(gv[pu + ".bias"], d(pg))
        ops.reduce_batch([(bw["dz"], None, gv[pd + ".bias"], 0)], ws["rscratch"])
        ops.outer_reduce(narrow=bw["dz"], wide=ws["G1"][i], scratch=ws["scratch"], out=gv[pd + ".weight"], M=M, C=C, L=Lt, transposed=0,
                         accumulate=0, wide_act=1)

    def _dvpt_bwd_scatter(self, ws, i, dG1, M):
        pd, pu, pg = self._dvpt_names(i)
        ops.skinny_up(lat=ws["dvb"]["dz"], w=self._d(pd + ".weight"), out=dG1, out_bf16=None if self.fp32 else ws["dG16"], gg_x=ws["G1"][i],
                      M=M, C=self.C, L=self.Lat, w_layout=1, accumulate=1)
        if self.fp32:
            ops.copy_(ws["dG16"], dG1)

    # ---- SSF (ssf.py): effective parameters per step, scale / shift gradients per site -----------------------------------------
    def _ssf_sites(self):
        """(scale name, shift name, kind, target) for every ssf_ada of the model, in forward order."""
        nm = self.names
        sites = [("ssf_scale_1", "ssf_shift_1", "linear", ("conv", "conv_proj.0.weight", "conv_proj.0.bias"))]
        for i in range(self.depth):
            a, m = nm.attn(i), nm.mlp(i)
            sites += [(a + ".ssf_scale_0", a + ".ssf_shift_0", "ln", (a + ".norm.weight", a + ".norm.bias")),
                      (a + ".ssf_scale_1", a + ".ssf_shift_1", "linear", (f"qkv{i}", a + ".to_qkv.weight", a + ".to_qkv.bias")),
                      (a + ".ssf_scale_2", a + ".ssf_shift_2", "linear", (f"out{i}", a + ".to_out.0.weight", a + ".to_out.0.bias")),
                      (m + ".ssf_scale_0", m + ".ssf_shift_0", "ln", (m + ".net.0.weight", m + ".net.0.bias")),
                      (m + ".ssf_scale_1", m + ".ssf_shift_1", "linear", (f"fc1{i}", m + ".net.1.weight", m + ".net.1.bias")),
                      (m + ".ssf_scale_2", m + ".ssf_shift_2", "linear", (f"fc2{i}", m + ".net.4.weight", m + ".net.4.bias"))]
        sites.append(("transformer.ssf_scale_1", "transformer.ssf_shift_1", "ln", ("transformer.norm.weight", "transformer.norm.bias")))
        return sites

    def _ssf_fold(self, train):
        """gamma' = gamma*s, beta' = beta*s + t;  W' = s[:,None]*W (operand dtype, + transpose for the dgrad), b' = b*s + t.
        Runs inside the recorded step: the scales and shifts are what the optimiser updates."""
        w, eff, raw = self._w16, self._eff, (lambda n: self.p[n].detach())
        for sn, tn, kind, tgt in self._ssf_sites():
            s_, t_ = raw(sn), raw(tn)
            if kind == "ln":
                gname, bname = tgt
                if gname not in eff:
                    eff[gname], eff[bname] = torch.empty_like(s_), torch.empty_like(s_)
                ops.ssf_fold_vec(raw(gname), s_, None, eff[gname])
                ops.ssf_fold_vec(raw(bname), s_, t_, eff[bname])
            else:
                key, wname, bname = tgt
                W = raw(wname)
                W2 = W.reshape(W.shape[0], -1)
                if key not in w:
                    w[key] = torch.empty(W2.shape, dtype=self.adt, device=W.device)
                    eff[bname] = torch.empty_like(s_)
                need_t = train and key != "conv"
                if need_t and key + "_t" not in w:
                    w[key + "_t"] = torch.empty((W2.shape[1], W2.shape[0]), dtype=self.adt, device=W.device)
                ops.ssf_fold_weight(W2, s_, w[key], w[key + "_t"] if need_t else None)
                ops.ssf_fold_vec(self.p[bname].detach() if bname in self.p else None, s_, t_, eff[bname])

    def _ssf_unfold(self, gv, bb, sites):
        """ScalingShiftingFeatures(freeze_vit=False): the backbone-gradient hooks see the EFFECTIVE tensors (W' = s o W, b' = b o s + t,
        gamma' = gamma o s, beta' = beta o s + t), so what they left in gv is dW', db', dgamma', dbeta'; the chain rule to the raw
        tensors is one multiplication by the site's scale (in place)."""
        if not bb:
            return
        for sn, tn, kind, tgt in sites:
            s_ = self.p[sn].detach()
            if kind == "ln":
                for n in tgt:
                    if n in bb:
                        ops.ssf_fold_vec(gv[n], s_, None, gv[n])
            else:
                _, wname, bname = tgt
                if wname in bb:
                    ops.ssf_fold_weight(gv[wname], s_, gv[wname], None)
                if bname in bb:
                    ops.ssf_fold_vec(gv[bname], s_, None, gv[bname])

    def _ssf_linear_grad(self, ws, gv, prefix, idx, dy, y0, M, N, y1=None, **kw):
        sn, tn = f"{prefix}.ssf_scale_{idx}", f"{prefix}.ssf_shift_{idx}"
        ops.ssf_colgrad(dy, y0, self.p[sn].detach(), self.p[tn].detach(), gv[sn], gv[tn], ws["ssf_scratch"], M, N, y1=y1, **kw)

    def _ssf_ln_grad(self, ws, gv, prefix, ln, dy, x, mean, rstd, M):
        C, tmp = self.C, ws["ssf_tmp"]
        ops.layernorm_bwd_affine(dy, x, mean, rstd, tmp[:C], tmp[C:], ws["scratch"], M, C)
        ops.ssf_ln_grad(tmp[:C], tmp[C:], self.p[prefix + ln + ".weight"].detach(), self.p[prefix + ln + ".bias"].detach(),
                        gv[prefix + ".ssf_scale_0"], gv[prefix + ".ssf_shift_0"])
