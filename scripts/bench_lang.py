#!/usr/bin/env python
"""Secondary metric (BASELINE config 3): LangPretrainer (PT-v3m1 + normalize + 3 criteria) fwd+bwd on B chunks of
102,400 Gaussians with 768-d targets, bf16 autocast, 1 GPU.  python scripts/bench_lang.py [B] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd import native as nv
from scenesplat_amd.pointcept_api import MODELS, RUNTIME
from scenesplat_amd.synthetic import LANG_PTV3, room_chunk

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
RUNTIME["attn_impl"] = nv.ATTN_MFMA; RUNTIME["conv_dtype"] = torch.bfloat16
crit = [dict(type="CosineSimilarity", reduction="mean", loss_weight=1.0), dict(type="L2Loss", reduction="mean", loss_weight=1.0),
        dict(type="AggregatedContrastiveLoss", temperature=0.2, reduction="mean", loss_weight=0.02, schedule="last_75")]
model = MODELS.build(dict(type="LangPretrainer", backbone=dict(type="PT-v3m1", **LANG_PTV3), criteria=crit)).cuda().train()
data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in room_chunk(256, 0, lang_dim=768, batch=B).items()}
data["epoch_progress"] = 0.5
n = data["feat"].shape[0]
side = torch.cuda.Stream()
plan = model.backbone.prepare_plan(data, stream=side)

def step():
    global plan
    model.zero_grad(set_to_none=True)
    p, plan = plan, None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model(dict(data, plan=p))["loss"]
    loss.backward()
    plan = model.backbone.prepare_plan(data, stream=side)
    return loss

for i in range(2):
    t = time.perf_counter(); l = step(); torch.cuda.synchronize()
    print(f"[lang] warmup {i}: {1e3*(time.perf_counter()-t):.1f} ms loss {l.item():.4f}", file=sys.stderr, flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): l = step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"metric": "Gaussians/s LangPretrainer fwd+bwd (config 3)", "value": n * steps / dt, "ms_per_step": dt / steps * 1e3,
                  "chunks": B, "gaussians": n, "loss": float(l), "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}), flush=True)
