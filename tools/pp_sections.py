"""Where post_process_mesh spends its time on a generated object's mesh (1 M triangles): the steps of gaussiananything_amd/mesh.py one by
one with a synchronisation in between.  usage (GPU box): python tools/pp_sections.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import _lib, mesh
import tools.tsdf_bench as tb  # noqa: F401  (builds nothing; reuse of its scene would need its main: a synthetic mesh instead)

dev = torch.device("cuda:0")
# a closed triangulated grid torus-like surface with ~1 M triangles plus floaters: the same operations at the same sizes
n = 708
idx = torch.arange(n * n, device=dev).view(n, n)
a, b, c, d = idx, idx.roll(-1, 0), idx.roll(-1, 1), idx.roll(-1, 0).roll(-1, 1)
t = torch.cat([torch.stack([a, b, c], -1).view(-1, 3), torch.stack([b, d, c], -1).view(-1, 3)]).int()
v = torch.rand(n * n, 3, device=dev)
nt, nv = t.shape[0], v.shape[0]
print("triangles", nt, "vertices", nv)
def T(label, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print(f"  {label}: {(time.perf_counter() - t0) * 1e3:.2f} ms"); return r
for rep in range(2):
    print("rep", rep)
    tl = T("long()", lambda: t.long())
    ea, eb = T("edge ends", lambda: (torch.cat([tl[:, 0], tl[:, 1], tl[:, 2]]), torch.cat([tl[:, 1], tl[:, 2], tl[:, 0]])))
    key, order = T("sort keys", lambda: torch.sort(torch.minimum(ea, eb) * nv + torch.maximum(ea, eb)))
    owner = T("owner", lambda: torch.arange(nt, device=dev).repeat(3)[order])
    pa, pb = T("pairs", lambda: (lambda same: (owner[:-1][same].contiguous(), owner[1:][same].contiguous()))(key[1:] == key[:-1]))
    label = torch.empty(nt, device=dev, dtype=torch.int32)
    T("union-find", lambda: _lib.check(_lib.lib().ga_mesh_cluster_labels(pa.data_ptr(), pb.data_ptr(), pa.numel(), label.data_ptr(), nt,
                                      ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "x"))
    cl = T("bincount", lambda: torch.bincount(label.long(), minlength=nt))
    sizes = T("sizes sort", lambda: torch.sort(cl[cl > 0]).values)
    keep = T("host read", lambda: max(int(sizes[-min(int(sizes.numel()), 10)]), 50))
    t2 = T("select", lambda: tl[cl[label.long()] >= keep])
    def compact():
        flags = torch.zeros(nv, dtype=torch.bool, device=dev); flags[t2.reshape(-1)] = True
        used = flags.nonzero().squeeze(1); remap = torch.cumsum(flags, 0) - 1
        t3 = remap[t2]
        return used, t3[(t3[:, 0] != t3[:, 1]) & (t3[:, 1] != t3[:, 2]) & (t3[:, 0] != t3[:, 2])]
    T("compact", compact)
    T("whole post_process_mesh", lambda: mesh.post_process_mesh(v, torch.zeros_like(v), t))
