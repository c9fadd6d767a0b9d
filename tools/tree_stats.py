#!/usr/bin/env python3
"""Per-height summary of the nested-dissection elimination tree the multifrontal Cholesky uses for one level
(host-only, through mgb_plan_create / mgb_plan_chol_tree).  usage: tree_stats.py [fem1d|fem2d|fem3d] L [level]"""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mgb_amd as M                      # noqa: E402
from mgb_amd import _lib                 # noqa: E402


def tree(kind, L, level=None, k=3):
    call, dptr, iptr, f64, i32 = _lib.call, _lib.dptr, _lib.iptr, _lib.f64, _lib.i32
    g = getattr(M, kind)(L) if kind != "fem3d" else M.fem3d(L, k)
    dim = {"fem1d": 1, "fem2d": 2, "fem3d": 3}[kind]
    x = f64(np.asarray(g.x).reshape(np.asarray(g.x).shape[0], -1))
    w = f64(g.w)
    Lv = len(g.refine)
    level = Lv - 1 if level is None else level
    h = C.c_void_p()
    call("mgb_geo_create", x.shape[0], x.shape[1], Lv, 1, dptr(x), dptr(w), C.byref(h))

    def put(name, S):
        S = sp.csr_matrix(S)
        S.sort_indices()
        rp, ci, va = i32(S.indptr), i32(S.indices), f64(S.data)
        call("mgb_geo_set_matrix", h, name.encode(), S.shape[0], S.shape[1], iptr(rp), iptr(ci), dptr(va))

    for kk, S in g.operators.items():
        put("op:" + kk, S)
    for kk, v in g.subspaces.items():
        for l, S in enumerate(v):
            put("sub:%s:%d" % (kk, l), S)
    state = M.DEFAULT_STATE if hasattr(M, "DEFAULT_STATE") else (("u", "dirichlet"), ("s", "full"))
    D = M.DEFAULT_D[dim]
    K = len(D)
    idx = list(range(K - dim - 1, K))
    iq = (C.c_int * (len(idx) - 1))(*idx[:-1])
    p = C.c_void_p()
    call("mgb_plan_create", h, len(state), _lib.str_array(state), K, _lib.str_array(D), len(idx) - 1, iq,
         idx[-1], level, C.byref(p))
    nn = C.c_int()
    call("mgb_plan_chol_tree", p, dim, 0, C.byref(nn), None, None, None)
    ns, nf, par = (np.zeros(nn.value, dtype=np.int32) for _ in range(3))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    call("mgb_plan_chol_tree", p, dim, nn.value, C.byref(nn), ip(ns), ip(nf), ip(par))
    call("mgb_plan_destroy", p)
    call("mgb_geo_destroy", h)
    height = np.zeros(nn.value, dtype=int)
    for t in range(nn.value):
        if par[t] >= 0:
            height[par[t]] = max(height[par[t]], height[t] + 1)
    return ns, nf, par, height


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "fem2d"
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    level = int(sys.argv[3]) if len(sys.argv) > 3 else None
    ns, nf, par, height = tree(kind, L, level)
    print("%s L=%d: N=%d, %d fronts, %d heights, front doubles %.3g" % (kind, L, ns.sum(), len(ns), height.max() + 1,
                                                                         float((nf.astype(float) ** 2).sum())))
    print("height count  ns(min/mean/max)   nf(min/mean/max)  panels  flops")
    for hh in range(height.max() + 1):
        m = height == hh
        a, b = ns[m].astype(float), nf[m].astype(float)
        fl = (a * b * b - a * a * b + a ** 3 / 3).sum()
        print("%5d %6d  %4d/%6.1f/%4d   %4d/%6.1f/%4d  %5d  %.3g" % (hh, m.sum(), a.min(), a.mean(), a.max(), b.min(),
                                                                      b.mean(), b.max(), -(-int(a.max()) // 32), fl))
