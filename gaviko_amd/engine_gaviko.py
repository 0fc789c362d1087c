"""GAViKO-specific half of the launch-plan engine: the MWSA (local window attention) chain and the GPA (gated prompt aggregation) chain,
forward and backward, on their own streams (gaviko.py:149-244).  Mixed into engine.Engine; every method only enqueues C-ABI launches."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch

from . import lib as L
from . import ops
from .engine_common import (SEED_EMB, SEED_LAYER, SEED_PROMPT, GRAPH_WARMUP, Names, PLAN_TIMING, SIDE_STREAM_PRIORITY, STEP_MODE, USE_GRAPHS, _ABLATE, _EPI_NAMES, _FIX_IN_LN, _LOC_SHIFT, _MODE, _SIDE_STREAMS, _on, evp_highpass_operator)  # noqa: F401


class GavikoPaths:
    # ---- GAViKO side paths --------------------------------------------------------------------------------------
    def _mwsa_fwd(self, ws, sv, i, si, lin, lout, gpa_local=False):
        """MWSA of layer i on the local stream (gaviko.py:229-244).  With self._fuse_next the up-projection kernel of layer i also runs layer
        i+1's entry (LayerNorm + proj_down + qkv of the rows it writes), so only layer 0 launches the entry kernel itself."""
        if not _on("noside"):
            return
        pre = f"transformer.local_attns.{i // self.share}"
        d, C, Lt, B = self._d, self.C, self.Lat, ws["B"]
        BN = B * self.N
        m = ws["mw"][si]
        chained = self._fuse_next and gpa_local and _on("loc_noupdown")
        if _on("loc_noupdown") and not (chained and i > 0):
            ops.skinny_down(x=lin, w=d(pre + ".proj_down.weight"), bias=d(pre + ".proj_down.bias"), ln_gamma=d(pre + ".norm.weight"),
                            ln_beta=d(pre + ".norm.bias"), mean=m["mean"], rstd=m["rstd"], y=m["lat"], w2=d(pre + ".qkv.weight"), y2=m["qkv"],
                            M=BN, C=C, L=Lt, L2=3 * Lt, act=0, w_layout=0, eps=1e-5)
        if _on("nowin"):
            ops.window_attn_fwd(qkv=m["qkv"], ctx=m["ctx"], lse=m["lse"], B=B, D=self.grid[0], H=self.grid[1], W=self.grid[2],
                                kd=self.win[0], kh=self.win[1], kw=self.win[2], L=Lt, scale=C ** -0.5, drop_p=sv["attn_drop"],
                                seed=2 * i, seed_ptr=ws["seed"])
        second = {}
        if gpa_local:                                       # ll = QuickGELU(proj_down(L')) (gaviko.py:156) of the rows this launch produces
            gpre, _ = self._gpa_names(i)
            g = ws["gp"][si]
            second = dict(w2=d(gpre + ".proj_down.0.weight"), bias2=d(gpre + ".proj_down.0.bias"), z2=g["zl"], y2=g["ll"], L2=Lt, act2=1)
        if chained and i + 1 < self.depth:                  # layer i+1's norm + proj_down + qkv of the same rows
            nx = f"transformer.local_attns.{(i + 1) // self.share}"
            mn = ws["mw"][si + 1 if sv["train"] else 0]
            second.update(nx_w=d(nx + ".proj_down.weight"), nx_bias=d(nx + ".proj_down.bias"), nx_ln_gamma=d(nx + ".norm.weight"),
                          nx_ln_beta=d(nx + ".norm.bias"), nx_mean=mn["mean"], nx_rstd=mn["rstd"], nx_lat=mn["lat"], nx_w2=d(nx + ".qkv.weight"),
                          nx_y2=mn["qkv"], nx_L2=3 * Lt, nx_eps=1e-5)
        if _on("loc_noupdown"):
            ops.skinny_up(lat=m["ctx"], w=d(pre + ".proj_up.weight"), bias=d(pre + ".proj_up.bias"), res=lin, out=lout, M=BN, C=C, L=Lt,
                          w_layout=0, drop_p=sv["proj_drop"], seed=2 * i + 1, seed_ptr=ws["seed"], **second)

    def _gpa_names(self, i):
        s = i // self.share
        pre = f"transformer.prompt_projs.{s}"
        ca, gl = pre + ".cls_analyzer.cls_analyzer_", pre + ".gl_balancer.gl_balancer_"
        return pre, dict(ca0_g=ca + ".0.weight", ca0_b=ca + ".0.bias", ca1_w=ca + ".1.weight", ca1_b=ca + ".1.bias", ca3_w=ca + ".3.weight",
                         ca3_b=ca + ".3.bias", gl0_g=gl + ".0.weight", gl0_b=gl + ".0.bias", gl1_w=gl + ".1.weight", gl1_b=gl + ".1.bias",
                         wgq=pre + ".global_attention.query_proj.weight", bgq=pre + ".global_attention.query_proj.bias",
                         wlq=pre + ".local_attention.query_proj.weight", blq=pre + ".local_attention.query_proj.bias")

    def _gpa_down_local(self, ws, i, si, lnew, B):
        """ll = QuickGELU(proj_down(L')) (gaviko.py:156): depends on the MWSA chain only, so it runs at its tail."""
        if "noside" in _ABLATE:
            return
        pre, _ = self._gpa_names(i)
        d, g = self._d, ws["gp"][si]
        ops.skinny_down(x=lnew, w=d(pre + ".proj_down.0.weight"), bias=d(pre + ".proj_down.0.bias"), z=g["zl"], y=g["ll"], M=B * self.N,
                        C=self.C, L=self.Lat, act=1, w_layout=0)

    def _gpa_fwd_latents(self, ws, i, si, g1, lnew, M, B, project, enh16=None):
        if "noside" in _ABLATE:
            return
        pre, names = self._gpa_names(i)
        d, C, Lt = self._d, self.C, self.Lat
        g = ws["gp"][si]
        if project:
            ops.skinny_down(x=g1, w=d(pre + ".proj_down.0.weight"), bias=d(pre + ".proj_down.0.bias"), z=g["zx"], y=g["xl"], M=M, C=C, L=Lt,
                            act=1, w_layout=0)
            self._gpa_down_local(ws, i, si, lnew, B)
        slot = {} if enh16 is None else dict(enh16=enh16, ld16=enh16.shape[-1], col16=self.mlp)
        ops.gpa_fwd(xl=g["xl"], ll=g["ll"], B=B, T=self.T, N=self.N, P=self.P, L=Lt, scale=Lt ** -0.5,
                    imp=g["imp"], gw=g["gw"], enh=g["enh"], prm=g["prm"], qg=g["qg"], ql=g["ql"], cg=g["cg"], cl=g["cl"],
                    lse_g=g["lse_g"], lse_l=g["lse_l"], **slot, **{k: d(v) for k, v in names.items()})

    def _gpa_fwd_up(self, ws, i, si, gout, M):
        pre, _ = self._gpa_names(i)
        d, g = self._d, ws["gp"][si]
        ops.skinny_up(lat=g["xl"], w=d(pre + ".proj_up.weight"), bias=d(pre + ".proj_up.bias"), out=gout, lat_override=g["enh"],
                      M=M, C=self.C, L=self.Lat, T=self.T, P=self.P, w_layout=0, accumulate=1)

    def _acc(self, i) -> int:
        """0 when layer i is the first (highest) layer of the sweep that touches its shared side-path module, else 1."""
        s = i // self.share
        top = min(self.depth - 1, s * self.share + self.share - 1)
        return 0 if i == top else 1

    def _gpa_bwd_core(self, ws, sv, gv, i, dGout, M, B, par, project=True):
        """Critical part of the GPA backward: dcomb = dGout . Wup and the latent-space backward -> dzx / dzl
        (what the main stream's dG1 update and the MWSA chain wait for)."""
        if "noside" in _ABLATE:
            return
        pre, names = self._gpa_names(i)
        d, C, Lt, P, T, N = self._d, self.C, self.Lat, self.P, self.T, self.N
        g, bw = ws["gp"][i], ws["bw"]
        if project:
            ops.skinny_down(x=dGout, w=d(pre + ".proj_up.weight"), y=bw["dcomb"], M=M, C=C, L=Lt, act=0, w_layout=1)
        ops.gpa_bwd(xl=g["xl"], ll=g["ll"], B=B, T=T, N=N, P=P, L=Lt, scale=Lt ** -0.5, imp=g["imp"], gw=g["gw"], enh=g["enh"], prm=g["prm"],
                    qg=g["qg"], ql=g["ql"], cg=g["cg"], cl=g["cl"], lse_g=g["lse_g"], lse_l=g["lse_l"], dcomb=bw["dcomb"], zx=g["zx"], zl=g["zl"],
                    dimp=bw["dimp"], dgw_part=bw["dgw_part"], dqg=bw["dqg"], dql=bw["dql"], dcg=bw["dcg"], dcl=bw["dcl"],
                    delta_g=bw["delta_g"], delta_l=bw["delta_l"], dprm=bw["dprm"], dcls=bw["dcls"], gate_partials=bw["gate_partials"],
                    dzx=bw["dzx"], dzl=bw["dzl"][par], **{k: d(v) for k, v in names.items()})

    def _gpa_bwd_params(self, ws, sv, gv, i, dGout, M, B, par):
        """Off the critical path: every parameter gradient of the GPA module (reads dGout, dzx, dzl, saved activations)."""
        if "noside" in _ABLATE or "noparams" in _ABLATE:
            return
        pre, names = self._gpa_names(i)
        d, C, Lt, P, T, N = self._d, self.C, self.Lat, self.P, self.T, self.N
        g, bw, sc = ws["gp"][i], ws["bw"], ws["scratch"]
        acc = self._acc(i)
        # gate parameters: one contiguous slice of the flat gradient buffer, in the kernel's order
        ng = ops.gpa_gate_param_count(Lt, P)
        first = gv[names["ca0_g"]]
        gate_flat = self._flat_grad["buf"][self._offset_of(names["ca0_g"]): self._offset_of(names["ca0_g"]) + ng]
        assert gate_flat.data_ptr() == first.data_ptr()
        gwd, gbd = gv[pre + ".proj_down.0.weight"], gv[pre + ".proj_down.0.bias"]
        BP = B * P
        dqg, dql, prm = bw["dqg"].view(BP, Lt), bw["dql"].view(BP, Lt), g["prm"].view(BP, Lt)
        small = [(bw["gate_partials"], None, gate_flat, acc),
                 (dqg, prm, gv[names["wgq"]], acc), (dqg, None, gv[names["bgq"]], acc),
                 (dql, prm, gv[names["wlq"]], acc), (dql, None, gv[names["blq"]], acc),
                 (bw["dzx"], None, gbd, acc, bw["dzl"][par])]                                 # proj_down bias: both token streams
        if self._pgrad:
            # ONE launch: proj_up (dWup = dGout^T . comb, dbup = colsum dGout; gaviko.py:187), proj_down fed by both token streams
            # (dWd = dzx^T . G1 + dzl^T . L'; gaviko.py:155-156) and the six small reductions; partial tiles summed by the last-arriving workgroup
            ops.param_grads(
                [dict(narrow=g["xl"], wide=dGout, lat_override=g["enh"], out=gv[pre + ".proj_up.weight"], colsum=gv[pre + ".proj_up.bias"], M=M, T=T, P=P,
                      transposed=1, accumulate=acc),
                 dict(narrow=bw["dzx"], wide=ws["G1"][i], narrow2=bw["dzl"][par], wide2=ws["Lc"][i + 1], out=gwd, M=M, M2=B * N, transposed=0,
                      accumulate=acc)],
                small, ws["pscratch"], ws["ptick"], C, Lt)
            return
        ops.outer_reduce(narrow=g["xl"], wide=dGout, lat_override=g["enh"], scratch=sc, out=gv[pre + ".proj_up.weight"],
                         colsum=gv[pre + ".proj_up.bias"], M=M, C=C, L=Lt, T=T, P=P, transposed=1, accumulate=acc)
        ops.reduce_batch(small, ws["rscratch"])
        # proj_down (shared by both streams): dWd = dzx^T.G1 + dzl^T.Lnew
        if M + B * N <= ops.OUTER_MAX_ROWS:                                   # both token streams in one pass
            ops.outer_reduce(narrow=bw["dzx"], wide=ws["G1"][i], narrow2=bw["dzl"][par], wide2=ws["Lc"][i + 1], scratch=sc, out=gwd, M=M, M2=B * N,
                             C=C, L=Lt, transposed=0, accumulate=acc)
        else:
            ops.outer_reduce(narrow=bw["dzx"], wide=ws["G1"][i], scratch=sc, out=gwd, M=M, C=C, L=Lt, transposed=0, accumulate=acc)
            ops.outer_reduce(narrow=bw["dzl"][par], wide=ws["Lc"][i + 1], scratch=sc, out=gwd, M=B * N, C=C, L=Lt, transposed=0, accumulate=1)

    def _gpa_bwd_scatter_g(self, ws, i, dG1, M):
        """main stream: dG1 += dzx . Wd, with the bf16 copy for the out-proj dgrad."""
        pre, _ = self._gpa_names(i)
        ops.skinny_up(lat=ws["bw"]["dzx"], w=self._d(pre + ".proj_down.0.weight"), out=dG1, out_bf16=None if self.fp32 else ws["dG16"],
                      M=M, C=self.C, L=self.Lat, w_layout=1, accumulate=1)
        if self.fp32:
            ops.copy_(ws["dG16"], dG1)

    def _mwsa_chain_bwd(self, ws, sv, gv, i, par, B, loc, after):
        """Local stream: dL += dzl . Wd (GPA's share), then the MWSA backward of layer i; starts once event `after` is reached.
        The last step of a layer's MWSA backward (dL_in = dL_out + LN'(dlat . Wd)) is deferred to the start of the next-lower layer's
        chain, where ONE kernel does it together with that layer's scatter and its first down-projection (self._fuse_bnd)."""
        self._ev_wait(loc, after)
        with torch.cuda.stream(loc):
            pend, fused = self._mwsa_pending, False
            if pend is not None and self._fuse_bnd and "noside" not in _ABLATE and "loc_noupdown" not in _ABLATE:
                self._mwsa_boundary(ws, sv, pend, i, par, B)
                fused = True
            else:
                if pend is not None:
                    self._mwsa_final(ws, pend, B)
                self._gpa_bwd_scatter_l(ws, i, ws["dL"][par], B, par)        # dL[par] was written on this stream
            self._scl_done = self._ev_record(loc)
            self._mwsa_bwd(ws, sv, gv, i, ws["dL"][par], ws["dL"][par ^ 1], B, have_dctx=fused, defer_final=True)
            self._mwsa_pending = i
            self._bucket_mark("loc", i)

    def _mwsa_flush(self, ws, B, loc, dead=False):
        """End of a backward plan: the deferred last step of the lowest layer of the sweep (dead: nobody reads its result -- skipped)."""
        if self._mwsa_pending is not None:
            if not dead:
                with torch.cuda.stream(loc):
                    self._mwsa_final(ws, self._mwsa_pending, B)
            self._mwsa_pending = None

    def _mwsa_final(self, ws, j, B):
        """dL_in = dL_out + LN'(dlat . Wd) of layer j (gaviko.py:231): the rank-L product never touches HBM."""
        if "noside" in _ABLATE or "loc_noupdown" in _ABLATE:
            return
        pre = f"transformer.local_attns.{j // self.share}"
        d, m, par = self._d, ws["mw"][j], (self.depth - 1 - j) & 1
        ops.skinny_up(lat=ws["bw"]["dlat"], w=d(pre + ".proj_down.weight"), res=ws["dL"][par], out=ws["dL"][par ^ 1], ln_x=ws["Lc"][j],
                      ln_mean=m["mean"], ln_rstd=m["rstd"], ln_gamma=d(pre + ".norm.weight"), M=B * self.N, C=self.C, L=self.Lat, w_layout=1)

    def _mwsa_boundary(self, ws, sv, j, i, par, B):
        """Layer j = i + 1's last step, layer i's GPA scatter and layer i's dctx = proj_drop'(dL) . Wup in one pass over the local-stream
        gradient (gvk_skinny_up with lat_b): dL[par] = dL[par ^ 1] + LN'(dlat_j . Wd_j) + dzl_i . Wd_gpa_i;  dctx_i = (dL[par] o mask_i) . Wup_i."""
        pj, pi_ = f"transformer.local_attns.{j // self.share}", f"transformer.local_attns.{i // self.share}"
        gpre, _ = self._gpa_names(i)
        d, m, bw = self._d, ws["mw"][j], ws["bw"]
        ops.skinny_up(lat=bw["dlat"], w=d(pj + ".proj_down.weight"), res=ws["dL"][par ^ 1], out=ws["dL"][par], ln_x=ws["Lc"][j],
                      ln_mean=m["mean"], ln_rstd=m["rstd"], ln_gamma=d(pj + ".norm.weight"), M=B * self.N, C=self.C, L=self.Lat, w_layout=1,
                      lat_b=bw["dzl"][par], w_b=d(gpre + ".proj_down.0.weight"),
                      w2=d(pi_ + ".proj_up.weight"), z2=bw["dctx"], L2=self.Lat, act2=0, w2_layout=1,
                      drop2_p=sv["proj_drop"], seed2=2 * i + 1, seed_ptr=ws["seed"])

    def _gpa_bwd_scatter_l(self, ws, i, dLnew, B, par):
        """MWSA chain: dL += dzl . Wd."""
        if "noside" in _ABLATE:
            return
        pre, _ = self._gpa_names(i)
        if _on("loc_noupdown"):
            ops.skinny_up(lat=ws["bw"]["dzl"][par], w=self._d(pre + ".proj_down.0.weight"), out=dLnew, M=B * self.N, C=self.C, L=self.Lat,
                          w_layout=1, accumulate=1)

    def _offset_of(self, name) -> int:
        return (self._flat_grad["views"][name].data_ptr() - self._flat_grad["buf"].data_ptr()) // 4

    def _mwsa_bwd(self, ws, sv, gv, i, dLout, dLin, B, have_dctx=False, defer_final=False):
        """MWSA backward of layer i on the local stream.  have_dctx: the layer-boundary kernel already produced dctx (_mwsa_boundary);
        defer_final: the last step (dL_in) is left to the next-lower layer's boundary kernel (_mwsa_chain_bwd).  The `_on(...)` guards are
        the timing ablations of DESIGN.md section 7b.3."""
        if not _on("noside"):
            return
        pre = f"transformer.local_attns.{i // self.share}"
        d, C, Lt = self._d, self.C, self.Lat
        BN = B * self.N
        m, bw, sc = ws["mw"][i], ws["bw"], ws["scratch_l"]
        lin = ws["Lc"][i]
        acc = self._acc(i)
        pd, seed_p, seed_a, sp = sv["proj_drop"], 2 * i + 1, 2 * i, ws["seed"]
        if _on("loc_noupdown") and not have_dctx:
            ops.skinny_down(x=dLout, w=d(pre + ".proj_up.weight"), y=bw["dctx"], M=BN, C=C, L=Lt, act=0, w_layout=1, drop_p=pd, seed=seed_p,
                            seed_ptr=sp)
        pgrad = self._pgrad and _on("loc_noouter") and _on("loc_nosmall") and (3 * Lt) % 4 == 0
        if _on("loc_noouter") and not pgrad:
            ops.outer_reduce(narrow=m["ctx"], wide=dLout, scratch=sc, out=gv[pre + ".proj_up.weight"], colsum=gv[pre + ".proj_up.bias"],
                             M=BN, C=C, L=Lt, transposed=1, accumulate=acc, drop_p=pd, seed=seed_p, seed_ptr=sp)
        if _on("nowin"):
            ops.window_attn_bwd(qkv=m["qkv"], ctx=m["ctx"], lse=m["lse"], dctx=bw["dctx"], delta=bw["wdelta"], dqkv=bw["dqkv"], B=B,
                                D=self.grid[0], H=self.grid[1], W=self.grid[2], kd=self.win[0], kh=self.win[1], kw=self.win[2], L=Lt,
                                scale=C ** -0.5, drop_p=sv["attn_drop"], seed=seed_a, seed_ptr=sp)
        if _on("loc_nosmall"):
            ops.skinny_down(x=bw["dqkv"], w=d(pre + ".qkv.weight"), y=bw["dlat"], M=BN, C=3 * Lt, L=Lt, act=0, w_layout=1)
        wd = d(pre + ".proj_down.weight")
        g_, b_ = d(pre + ".norm.weight"), d(pre + ".norm.bias")
        if pgrad:
            # ONE launch at the tail of the layer's chain (dLout stays untouched until the layer after next rewrites its buffer): proj_up behind
            # proj_drop (gaviko.py:242-243), LayerNorm + proj_down through Q = dlat^T . xhat and S = colsum dlat (gaviko.py:231-232), the qkv matrix
            ops.param_grads(
                [dict(narrow=m["ctx"], wide=dLout, out=gv[pre + ".proj_up.weight"], colsum=gv[pre + ".proj_up.bias"], M=BN, transposed=1, accumulate=acc,
                      drop_p=pd, seed=seed_p),
                 dict(narrow=bw["dlat"], wide=lin, mean=m["mean"], rstd=m["rstd"], out=gv[pre + ".proj_down.weight"], aff_w=wd, aff_gamma=g_, aff_beta=b_,
                      aff_dgamma=gv[pre + ".norm.weight"], aff_dbeta=gv[pre + ".norm.bias"], aff_dbias=gv[pre + ".proj_down.bias"], M=BN, accumulate=acc),
                 # the qkv matrix (gaviko.py:205,233): dWqkv[c][l] = sum_m dqkv[m][c] lat[m][l] -- the same outer product with a 3L-wide "wide"
                 dict(narrow=m["lat"], wide=bw["dqkv"], out=gv[pre + ".qkv.weight"], M=BN, C=3 * Lt, transposed=1, accumulate=acc)],
                [], ws["pscratch_l"], ws["ptick_l"], C, Lt, seed_ptr=sp)
        if _on("loc_noouter") and not pgrad:
            # Q[l][c] = sum_m dlat[m][l] xhat[m][c], S[l] = sum_m dlat[m][l]  ->  dWd, dbd, dgamma, dbeta in one tiny kernel
            ops.outer_reduce(narrow=bw["dlat"], wide=lin, mean=m["mean"], rstd=m["rstd"], scratch=sc, out=bw["Q"], M=BN, C=C, L=Lt,
                             transposed=0, accumulate=0)
        if _on("loc_nosmall") and not pgrad:
            # qkv weight gradient (dqkv^T . lat) and S[l] = sum_m dlat[m][l] in one two-stage reduction
            ops.reduce_batch([(bw["dqkv"], m["lat"], gv[pre + ".qkv.weight"], acc), (bw["dlat"], None, bw["S"], 0)], ws["rscratch_l"])
            ops.ln_lowrank_affine(bw["Q"], bw["S"], wd, g_, b_, gv[pre + ".proj_down.weight"], gv[pre + ".norm.weight"],
                                  gv[pre + ".norm.bias"], gv[pre + ".proj_down.bias"], Lt, C, accumulate=bool(acc))
        if _on("loc_noupdown") and not defer_final:
            # dL_in = dL_out + LN'(dlat . Wd): the rank-L product never touches HBM
            ops.skinny_up(lat=bw["dlat"], w=wd, res=dLout, out=dLin, ln_x=lin, ln_mean=m["mean"], ln_rstd=m["rstd"], ln_gamma=g_, M=BN, C=C,
                          L=Lt, w_layout=1)
