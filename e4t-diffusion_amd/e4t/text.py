"""CLIP text encoder with ``inputs_embeds`` on the HIP kernels — SURVEY.md §8f row N3 (reference:
e4t/models/modeling_clip.py:9-82 patches transformers' CLIPTextModel so the E4T domain embedding can be written into the
token embeddings; pretrain_e4t.py:616,630-634 run it between the two UNet passes, forward + gradient w.r.t. the embeddings).

Module tree / parameter names from ``checkpoint_trees.py`` (= the HF checkpoint keys), so real weights load by key.  Frozen weights (both reference scripts by default): a fused q|k|v weight per layer is built once and only dX
is propagated: LayerNorm (residual gradient folded in) -> one (tokens x 3w) GEMM -> causal fused attention -> out-proj GEMM with
the residual in its epilogue -> LayerNorm -> fc1 -> quick_gelu / gelu -> fc2 (+ residual).  Trainable weights
(tuning_e4t.py --train_text_encoder, :145-146): the same kernels; the fused q|k|v weight is re-assembled every forward
as a differentiable torch.cat of the three parameters, so the TN weight-gradient GEMM's result is split back onto them by
autograd, and the LayerNorm / bias gradients come from the kernels' parameter-gradient outputs.  There is no other path.

Launch-bound: at B x 77 tokens every kernel of the 12 (23) layers runs 5-25 us while the Python side of an op costs ~30 us, so in the
training step the GPU idled ~2.5 ms per step in front of these launches (tools/idle_report.py).  With frozen weights the layer
stack is therefore captured once per input shape into two HIP graphs (forward | backward w.r.t. the embeddings,
torch.cuda.make_graphed_callables) and replayed: same kernels, same order, bit-identical results, ~250 launches -> 2 per step.
It is off under E4T_LAUNCH_LOG (the per-launch log the roofline tools join with
rocprofv3's trace only sees launches that go through the host)."""
from __future__ import annotations

import os
import warnings

import torch
from torch import nn

from . import functional as Fn
from . import ops
from .checkpoint_trees import CLIPTextModel as _CLIPTextTree


class CLIPTextModel(_CLIPTextTree):
    def __init__(self, **cfg):
        super().__init__(**cfg)
        self._fused = None
        self._graphs = {}            # (shape, dtype) -> (fused-weights key, graphed callable)
        self._graph_ok = not os.environ.get("E4T_LAUNCH_LOG")

    def _prepare(self, trainable):
        """Per layer: fused q|k|v weight + bias and the PreparedLinear handles of every projection.  Frozen: detached copies made
        once (re-made when a checkpoint load or resize replaced / rewrote the parameters).  Trainable: a differentiable cat,
        every forward (the bf16 compute copy has to be re-cast after each optimiser step anyway)."""
        layers = self.text_model.encoder.layers
        if not trainable:
            key = tuple((l.self_attn.q_proj.weight.data_ptr(), l.self_attn.q_proj.weight._version, l.self_attn.v_proj.weight._version) for l in layers)
            if self._fused is not None and self._fused[0] == key:
                return self._fused[1]
        out = []
        for i, l in enumerate(layers):
            a = l.self_attn
            w = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], dim=0)
            b = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], dim=0)
            if not trainable:
                w, b = nn.Parameter(w.detach(), requires_grad=False), nn.Parameter(b.detach(), requires_grad=False)
            keep = self._fused[1][i] if (trainable and self._fused is not None and len(self._fused[1]) == len(layers)) else None
            out.append(dict(wqkv=w, bqkv=b, pqkv=Fn.PreparedLinear(w),
                            pout=keep["pout"] if keep else Fn.PreparedLinear(a.out_proj.weight),
                            pfc1=keep["pfc1"] if keep else Fn.PreparedLinear(l.mlp.fc1.weight),
                            pfc2=keep["pfc2"] if keep else Fn.PreparedLinear(l.mlp.fc2.weight)))
        self._fused = (None if trainable else key, out)
        return out

    def forward(self, input_ids=None, inputs_embeds=None):
        tm = self.text_model
        trainable = torch.is_grad_enabled() and any(p.requires_grad for p in tm.encoder.parameters())
        if inputs_embeds is None:
            inputs_embeds = tm.embeddings.token_embedding(input_ids)
        if (self._graph_ok and not trainable and inputs_embeds.is_cuda and torch.is_grad_enabled() and inputs_embeds.requires_grad
                and not torch.cuda.is_current_stream_capturing()):
            return (self._replay(inputs_embeds),)
        return (self._encode(inputs_embeds, trainable),)

    def _replay(self, inputs_embeds):
        """The frozen layer stack as a forward and a backward HIP graph per input shape; re-captured when the weights were
        reloaded (the graphs read the fused q|k|v copies `_prepare` made).  Outputs are copied out of the graphs' static
        buffers: the UNet's cross-attention keeps the context until its own backward."""
        self._prepare(False)
        sig = (tuple(inputs_embeds.shape), inputs_embeds.dtype)
        hit = self._graphs.get(sig)
        if hit is None or hit[0] != self._fused[0]:
            sample = torch.zeros_like(inputs_embeds).requires_grad_(True)
            be = ops.backend()
            ws_before = set(getattr(be, "_ws", {}))
            try:
                # ops.capture_guard: the data loader's prefetch worker is paused (an allocation, copy or event wait from another host
                # thread invalidates a capture in global error mode) and the garbage collector is off (destroying an old graph during
                # a capture aborts the process).  Call `prepare_graphs` before the training loop to keep the pause out of the first step.
                with ops.capture_guard():
                    torch.cuda.synchronize()
                    fn = torch.cuda.make_graphed_callables(lambda e: self._encode(e, False), (sample,))
                # scratch buffers the warm-up / capture streams made the backend allocate (64 MB each, keyed by stream): the graph's
                # own copy lives in its memory pool, the host-side handles are dead weight
                for k in [k for k in getattr(be, "_ws", {}) if k not in ws_before]:
                    del be._ws[k]
            except Exception as ex:          # capture unsupported in this process: run the launches from the host as before
                warnings.warn(f"CLIP text encoder: HIP graph capture failed ({type(ex).__name__}: {ex}); running eagerly")
                self._graph_ok = False
                return self._encode(inputs_embeds, False)
            hit = self._graphs[sig] = (self._fused[0], fn)
        return hit[1](inputs_embeds).clone()

    def prepare_graphs(self, batch_size, device, requires_grad=True):
        """Capture the frozen stack's graphs for (batch_size, max_len, width) inputs NOW — before the training loop and its loader
        thread start — instead of inside the first step.  No-op when graphs are off, the weights train, or on the CPU."""
        dev = torch.device(device)
        if not (self._graph_ok and dev.type == "cuda") or any(p.requires_grad for p in self.text_model.encoder.parameters()):
            return False
        cfg = self.config
        e = torch.zeros((batch_size, cfg["max_len"], cfg["hidden_size"]), device=dev, requires_grad=requires_grad)
        with torch.enable_grad():
            self._replay(e)
        return self._graph_ok

    def _encode(self, inputs_embeds, trainable):
        tm = self.text_model
        B, S, W = inputs_embeds.shape
        cfg = self.config
        H = cfg["num_heads"]
        DH = W // H
        act = Fn.quick_gelu if cfg["act"] == "quick_gelu" else Fn.gelu
        x = (inputs_embeds + tm.embeddings.position_embedding.weight[:S]).to(ops.ACT).reshape(B * S, W).contiguous()
        for l, f in zip(tm.encoder.layers, self._prepare(trainable)):
            n, xs = Fn.layer_norm_skip(x, l.layer_norm1.weight, l.layer_norm1.bias, l.layer_norm1.eps)
            qkv = Fn.linear(n, f["wqkv"], f["bqkv"], f["pqkv"])
            a = Fn.attention(qkv, None, B, H, S, S, DH, DH ** -0.5, causal=True)
            x = Fn.linear(a, l.self_attn.out_proj.weight, l.self_attn.out_proj.bias, f["pout"], residual=xs)
            n, xs = Fn.layer_norm_skip(x, l.layer_norm2.weight, l.layer_norm2.bias, l.layer_norm2.eps)
            h = act(Fn.linear(n, l.mlp.fc1.weight, l.mlp.fc1.bias, f["pfc1"]))
            x = Fn.linear(h, l.mlp.fc2.weight, l.mlp.fc2.bias, f["pfc2"], residual=xs)
        fl = tm.final_layer_norm
        y = Fn.layer_norm(x, fl.weight, fl.bias, fl.eps)
        return y.view(B, S, W)
