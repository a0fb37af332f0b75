import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
import torch
from bc_workloads import harness
model = harness.build_model("resnet18", block_policy="fixed", block_size=128, block_target=0.5, device="cuda", channels_last=True, block_graph=1)
clips = [harness.synthetic_clip(20, (1, 3, 1024, 2048), seed=0, device="cuda")]
for _ in range(2):
    harness.run_clip(model, clips[0])
print("resident", harness.measure_fps(model, clips, 3, 1)[0])
h = [[f.cpu() for f in clips[0]]]
for cc in (False, True, False, True):
    print("cross_clip", cc, harness.measure_fps_with_upload(model, h, n_clips=3, warmup_clips=1, prefetch=True, cross_clip=cc)[0], flush=True)
print("n_clips 6, cross", harness.measure_fps_with_upload(model, h, n_clips=6, warmup_clips=2, prefetch=True, cross_clip=True)[0])
print("n_clips 6, no cross", harness.measure_fps_with_upload(model, h, n_clips=6, warmup_clips=2, prefetch=True, cross_clip=False)[0])
