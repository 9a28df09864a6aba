"""Measurement aid: per-wave wall-clock stamps of the blend kernel (build: tools/blend_variants.sh stamps "-DGA_BLEND_STAMPS=20000000").
usage (GPU box): python tools/blend_stamps.py [surface|stress]"""
import os, shutil, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MAIN = os.path.join(ROOT, "gaussiananything_amd", "lib", "libga_mi355.so")
shutil.copy(MAIN, MAIN + ".bak")
shutil.copy(os.path.join(ROOT, "tools", "_build", "libga_%s.so" % (sys.argv[2] if len(sys.argv) > 2 else "stamps")), MAIN)
try:
    from gaussiananything_amd import synthetic
    from gaussiananything_amd.diff_surfel_rasterization import SurfelForwardPlan
    OFF = 20_000_000
    dev = torch.device("cuda:0")
    scene = sys.argv[1] if len(sys.argv) > 1 else "surface"
    cams = synthetic.eval_cameras(8)
    g = synthetic.surface_surfels(100000, seed=1)[0] if scene == "surface" else synthetic.random_surfels(100000, seed=0)[0]
    m, o, s, r, c = [t.to(dev) for t in synthetic.split_gaussians(g)]
    plan = SurfelForwardPlan(m, o, c, s, r, cams["cam_view"].to(dev), cams["cam_view_proj"].to(dev), torch.ones(3, device=dev), 512, 512)
    plan.run(); plan.ensure_capacity()
    for _ in range(3): plan.run()
    torch.cuda.synchronize()
    ws = plan.ws
    cap = ws.capacity
    nwords = (cap // 256 + 1) * 15 * 256
    scr = ws.section("seg_scratch", torch.int64, nwords)
    nt = 8 * 1024
    seg_region = min(cap // 256, 512)
    nwg = seg_region + nt
    scr[OFF:OFF + nwg * 16].zero_()
    plan.run(); torch.cuda.synchronize()
    st = scr[OFF:OFF + nwg * 16].cpu().numpy().reshape(nwg, 4, 4)
    order = ws.section("tile_order", torch.int32, nt * 4).cpu().numpy().reshape(nt, 4)
    status = ws.status().cpu().numpy()
    nlong = int(status[7])
    t0 = st[..., 0][st[..., 0] > 0].min()
    ent, ext, rdy = (st[..., 0] - t0) / 100.0, (st[..., 1] - t0) / 100.0, (st[..., 3] - t0) / 100.0   # microseconds
    live = st[..., 0] > 0
    print("kernel span %.1f us; waves stamped %d" % (ext[live].max(), live.sum()))
    # resident consumer waves over time
    grid = np.arange(0, ext[live].max() + 1, 2.0)
    cons = live.copy(); pass
    occ = [(np.sum((ent[cons] <= t) & (ext[cons] > t))) for t in grid]
    busy = [(np.sum((rdy[cons] <= t) & (ext[cons] > t) & (st[..., 3][cons] > 0))) for t in grid]
    print("time us      :", " ".join("%5d" % t for t in grid[::5]))
    print("resident cons:", " ".join("%5d" % v for v in occ[::5]))
    print("past 1st wait:", " ".join("%5d" % v for v in busy[::5]))
    n = np.zeros(nwg, int); n[seg_region:] = 0
    pos = np.arange(nwg) - seg_region + nlong
    ok = (np.arange(nwg) >= seg_region) & (pos < nt)
    n[ok] = order[pos[ok], 2]
    wg_live = live[:, 0]
    dur = ext[:, :4].max(1) - ent[:, 0]
    start_lat = np.where(st[:, :4, 3] > 0, rdy[:, :4] - ent[:, :4], np.nan)
    for lo, hi in ((0, 0), (1, 64), (65, 256), (257, 512), (513, 1023), (1024, 2047)):
        sel = ok & wg_live & (n >= lo) & (n <= hi)
        if sel.sum():
            cw = ext[sel][:, :4] - ent[sel][:, :4]
            print("tiles with %4d..%4d entries: %5d WGs, WG lifetime mean %.1f us (max %.1f), consumer lifetime mean %.1f, first-chunk wait mean %.1f us, producer lifetime %.1f"
                  % (lo, hi, sel.sum(), dur[sel].mean(), dur[sel].max(), cw.mean(), np.nanmean(start_lat[sel]) if np.isfinite(start_lat[sel]).any() else 0, 0.0))
    segw = wg_live & (np.arange(nwg) < seg_region)
    if segw.sum():
        print("segment WGs: %d, lifetime mean %.1f max %.1f us" % (segw.sum(), dur[segw].mean(), dur[segw].max()))
    print("sum of consumer wave lifetimes %.0f us; /kernel span = %.1f waves resident on average (of %d slots)" % (
        (ext[cons] - ent[cons]).sum(), (ext[cons] - ent[cons]).sum() / ext[live].max(), 256 * 12))
    tot_wait = np.nansum(start_lat)
    print("sum of first-chunk waits %.0f us = %.1f%% of consumer lifetime" % (tot_wait, 100 * tot_wait / (ext[cons] - ent[cons]).sum()))
finally:
    shutil.copy(MAIN + ".bak", MAIN); os.remove(MAIN + ".bak")
