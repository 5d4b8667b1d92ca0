"""What the setup and the finish of the headline launch consist of (8 192 MPC QPs): the whole launch timed with HIP events under
parameter sets that switch one part off at a time.  max_iter = 1 ends every item with MAX_ITERATIONS after setup + one iteration
(no polish: report only); scaling off removes the Ruiz passes; polish off removes the second factorisation + refinement."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
import bench, smooth_feedback_amd as sfb

dev = torch.device("cuda:0")
wl = bench.MPCWorkload(sfb, 0, dev)
st = torch.cuda.Stream()


def timed(label, **kw):
    wl.prm = sfb.QPSolverParams(**kw)
    with torch.cuda.stream(st):
        for _ in range(2): wl.step(st)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): wl.step(st)
        e1.record(st); st.synchronize()
    it = wl.out[0].double().mean().item()
    print("%-46s %8.3f ms   mean iterations %.1f" % (label, e0.elapsed_time(e1) / 5, it))


timed("defaults")
timed("polish off", polish=False)
timed("polish_iter 0 (second factorisation, no refinement)", polish_iter=0)
timed("polish_iter 1", polish_iter=1)
timed("max_iter 1 (setup + 1 iteration + report)", max_iter=1)
timed("max_iter 1, scaling off", max_iter=1, scaling=False)
timed("max_iter 2", max_iter=2)
timed("max_iter 11", max_iter=11)
