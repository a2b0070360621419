"""Phase clock of fem_ball_newton_kernel (library variant built with -DTACEX_BALL_CLOCK, TACEX_LIB_TAG=bclk): env 0 / the last env print their
cycle counts per phase at the end of every step."""
import sys, torch
from tacex_amd.uipc.gelpad_scene import FemBallScene
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = FemBallScene(B, "cuda:0", max_newton_iter=64)
for i in range(14):
    sc.step(i); torch.cuda.synchronize()
    print(f"-- step {i}: newton mean {float(sc.sim.step_info[:,0].mean()):.2f}", flush=True)
