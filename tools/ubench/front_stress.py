"""Stress of the front launch's parity statement (tests/test_front_gpu.py::test_front_launch_equals_act_step_then_guarded_learn) over many shapes:
env counts that are not multiples of the acting workgroup's 32 rows, rings barely twice the env count, both acting formats, both update sequences.
    python3 tools/ubench/front_stress.py [rounds]"""
import itertools
import sys

sys.path.insert(0, ".")
from hirl4ucav_amd.agents import engine  # noqa: E402
from tests import test_front_gpu as T  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fn = T.test_front_launch_equals_act_step_then_guarded_learn.__wrapped__ if hasattr(T.test_front_launch_equals_act_step_then_guarded_learn, "__wrapped__") else T.test_front_launch_equals_act_step_then_guarded_learn
k = 0
for r in range(rounds):
    for n, act, staged, use_bc in itertools.product((96, 500, 1000, 2048, 4096, 8192), ("f32", "f32x9", "bf16"), (False, True), (True, False)):
        cap = max(2 * n + 37 + 100 * r, 512)
        try:
            fn(engine, use_bc, 0.0 if use_bc else 0.01, n, cap, act, staged)
            ok = "ok"
        except AssertionError as e:
            msg = str(e).splitlines()[0] if str(e) else repr(e)
            ok = "FAIL " + msg[:200]
        k += 1
        print(f"n={n} cap={cap} act={act} staged={staged} use_bc={use_bc}: {ok}", flush=True)
print("cases", k)
