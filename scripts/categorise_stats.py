"""Buckets a steady_state_stats.py CSV by kernel family (MIOpen igemm, Tensile GEMM, ATen element-wise, ours, ...).
usage: categorise_stats.py <steady.csv> [rows]"""
import csv
import sys

rows = list(csv.reader(open(sys.argv[1])))[2:]


def family(n):
    for key, name in (("igemm_fwd", "miopen igemm fwd"), ("igemm_bwd", "miopen igemm bwd"), ("igemm_wrw", "miopen igemm wrw"),
                      ("Cijk", "tensile gemm"), ("_ZN2ck", "ck conv/gemm"), ("naive_conv", "miopen naive conv"),
                      ("BatchNorm", "batchnorm"), ("batch_norm", "batchnorm"), ("dd::", "dd:: (ours)"), ("direct_copy", "layout/dtype copy"),
                      ("CatArray", "cat"), ("depthwise", "aten depthwise"), ("reduce_kernel", "aten reduce"),
                      ("SubTensorOp", "zero/fill"), ("fillBuffer", "zero/fill"), ("FillFunctor", "zero/fill"),
                      ("layer_norm", "layernorm"), ("LayerNorm", "layernorm"), ("multi_tensor", "optimizer"),
                      ("elementwise", "aten element-wise"), ("upsample", "aten upsample")):
        if key in n:
            return name
    return "other: " + n[:60]


cats = {}
for r in rows:
    a = cats.setdefault(family(r[0]), [0.0, 0.0])
    a[0] += float(r[1]); a[1] += float(r[2])
tot = sum(v[1] for v in cats.values())
print("kernel time %.3f ms/step, %d dispatches/step" % (tot / 1e6, sum(v[0] for v in cats.values())))
for k, v in sorted(cats.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print("%-62s %7.1f calls %8.3f ms %5.1f%%" % (k, v[0], v[1] / 1e6, 100 * v[1] / tot))
