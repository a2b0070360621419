"""Why some envs of the ball scene need 6-9 Newton iterations at the default tolerances: library variant built with -DTACEX_BALL_TRACE
(TACEX_LIB_TAG=btrace) prints iterations >= 3 of every env; this script steps the bench scene and tags the output with the step index."""
import sys, torch
from tacex_amd.uipc.gelpad_scene import FemBallScene
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = FemBallScene(B, "cuda:0", max_newton_iter=64)
for i in range(30):
    sc.step(i); torch.cuda.synchronize()
    si = sc.sim.step_info
    print(f"-- step {i}: newton mean {float(si[:,0].mean()):.2f} max {int(si[:,0].max())} (env {int(si[:,0].argmax())})", flush=True)
