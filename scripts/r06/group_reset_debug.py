"""Debug of test_group_member_reset_alone_gets_a_fresh_evaluation: which envs / markers of member 1 differ, and the trajectory state."""
import sys, torch
sys.path.insert(0, "tests")
from pathlib import Path
from test_sensor_gpu import make_sensor
from tacex_amd import GelSightSensorGroup
from tacex_amd.calibration import CALIB_GELSIGHT_MINI
from tacex_amd.utils.synthetic import synthetic_depth_maps

calib = CALIB_GELSIGHT_MINI
n = 3
ref = [make_sensor(calib, n), make_sensor(calib, n)]
for s in ref:
    s.initialize()
mem = [make_sensor(calib, n), make_sensor(calib, n)]
GelSightSensorGroup(mem)
hm = [synthetic_depth_maps(n, 240, 320, seed=41 + k, flat_fraction=0.0)[0].cuda() / 1000.0 for k in range(2)]
def ts(s):
    return s.marker_motion_simulator._traj_state.clone()
for sensors in (ref, mem):
    for k, s in enumerate(sensors):
        s.set_camera_depth(hm[k])
        s.update(dt=0.01, force_recompute=True)
print("after step 0: traj equal", [torch.equal(ts(ref[k]), ts(mem[k])) for k in range(2)], "markers equal",
      [torch.equal(ref[k].data.output["marker_motion"], mem[k].data.output["marker_motion"]) for k in range(2)])
ref[1].reset([1]); mem[1].reset([1])
print("after reset: traj equal", [torch.equal(ts(ref[k]), ts(mem[k])) for k in range(2)])
print(" ref traj m1", ts(ref[1]).cpu().numpy().round(4).tolist()); print(" mem traj m1", ts(mem[1]).cpu().numpy().round(4).tolist())
for sensors in (ref, mem):
    for k in (0, 1):
        sensors[k].update(dt=0.01)
for k in (0, 1):
    a, b = ref[k].data.output, mem[k].data.output
    d = (a["marker_motion"] - b["marker_motion"]).abs().amax(dim=(1, 2, 3))
    print("member", k, "marker diff per env", d.cpu().tolist(), "indent", ref[k].indentation_depth.cpu().tolist(), mem[k].indentation_depth.cpu().tolist())
    print(" traj equal", torch.equal(ts(ref[k]), ts(mem[k])))
    print(" ref", ts(ref[k]).cpu().numpy().round(4).tolist()); print(" mem", ts(mem[k]).cpu().numpy().round(4).tolist())
