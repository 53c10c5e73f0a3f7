import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from dt4image_restoration_amd import synthetic, weights, data as D
from dt4image_restoration_amd.denoiser import UNetDenoiser2D
from dt4image_restoration_amd.env import PnPEnv
from dt4image_restoration_amd.policy import DecisionTransformer, DecisionTransformerConfig
from dt4image_restoration_amd.drivers.greedy import GreedyEvaluator
n = 64
den = UNetDenoiser2D.seeded(0, "unit_gain")
m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm")).cuda()
m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-8.0, head_gain=1.0))
prob = synthetic.make_problem(n, 256, 256, accel=4.0, seed=1234)
mat = {k: torch.from_numpy(np.asarray(v)) for k, v in prob.items()}
ev = GreedyEvaluator(m, PnPEnv(30, den, "cuda"), max_timesteps=30, device_type="cuda")
rtg = torch.full((n,), D.normalised_rtg(10.0)); task = torch.full((n,), 4)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ev.run(mat, rtg, task)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"greedy rollout {n}x256x256, {int(r.stop_time.max())} steps: {dt*1e3:.1f} ms total, {dt*1e3/max(int(r.stop_time.max()),1):.2f} ms/step, mean PSNR {float(r.reward.mean()):.2f}")

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    r = ev.run(mat, rtg, task)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
