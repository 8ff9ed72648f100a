"""Developer tool: how long the parts of a split object wait for each other (a -DM3T_PHASE_TIMING -DM3T_EXCHANGE_STAMPS_ONLY
build: the exchange stamps without the phase marks).  Per correspondence iteration and part of object 0, last frame:
publish = stamp(published) - stamp(publish start), wait = stamp(collected) - stamp(published).  (s_memtime bases
differ between CUs: only differences of one part's own stamps mean anything.)
  python tools/exchange_waits.py <lib> [objects]"""
import ctypes as C, importlib, os, sys
os.environ.setdefault("M3T_INPUT_WORKERS", "auto")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("3dobjecttracking_amd")
import scenes
lib, n_obj = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64
hip = pkg.CApi(lib, "m3t_hip_")
inputs = scenes.Inputs(n_obj, 8, n_divides=4, n_models=min(8, n_obj))
inst = scenes.Instance(hip, inputs)
inst.upload_frame(0)
inst.tracker.StartModalities(0)
g = hip.lib.m3t_hip_debug_exchange_times
g.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
shape = (C.c_int * 4)()
for k in range(1, 8):
    inst.upload_frame(k); inst.tracker.ExecuteTrackingStep(k)
    xt = (C.c_ulonglong * 768)()
    g(hip.ctx, xt)
    hip.call("get_step_shape", shape)
    parts = shape[1]
    if k < 5:
        continue
    print("frame %d, %d parts" % (k, parts))
    for rnd in range(7):
        start = [xt[(0 * 16 + rnd) * 16 + p] for p in range(parts)]
        pub = [xt[(1 * 16 + rnd) * 16 + p] for p in range(parts)]
        done = [xt[(2 * 16 + rnd) * 16 + p] for p in range(parts)]
        print("  search %d: publish %s  wait %s" % (rnd, [int(b - a) for a, b in zip(start, pub)], [int(c - b) for b, c in zip(pub, done)]))
    # between searches: collected(rnd) -> publish start(rnd + 1) of the same part = the Newton steps + the next search's own work
    for rnd in range(6):
        print("  search %d -> %d, collected -> next publish start: %s" % (rnd, rnd + 1, [int(xt[(0 * 16 + rnd + 1) * 16 + p] - xt[(2 * 16 + rnd) * 16 + p]) for p in range(parts)]))
