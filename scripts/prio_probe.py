"""Experiment: HIP stream priorities for the main / weight-gradient streams (no effect measured: 13.13-13.21 ms/step)."""
import sys, io, time, torch
sys.path.insert(0, '/root/repo')
from pqdet_amd import optim
from pqdet_amd.dataset import labels
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel
print('priority range', torch.cuda.Stream.priority_range())
def run(main_prio, side_prio):
    torch.manual_seed(0)
    model = DetectionModel(io.StringIO(zoo.cfg_text('mobilenetv2-fpn'))).cuda(); model.train()
    opt = optim.Adam(model.parameters(), lr=4e-4)
    x = torch.randn(16, 3, 512, 512).cuda(); tgt = tuple(t.cuda() for t in labels.synthetic_targets(16, 512, 20, seed=1))
    model._side = torch.cuda.Stream(priority=side_prio)
    main = torch.cuda.Stream(priority=main_prio)
    main.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(main):
        def step():
            opt.zero_grad(); out = model(x, tgt); out['loss'].mean().backward(); opt.step()
        for _ in range(10): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): step()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('main prio %d side prio %d: %.3f ms/step' % (main_prio, side_prio, dt / 40 * 1e3), flush=True)
lo, hi = torch.cuda.Stream.priority_range()
run(0, 0); run(hi, 0); run(hi, lo) if lo != 0 else None; run(0, hi)
