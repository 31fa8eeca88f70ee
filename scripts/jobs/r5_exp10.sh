set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5x10; mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
DD_PROBE_ROWS=90 DD_PROBE_FILTER="aten::add,aten::copy_,aten::mul,aten::addcmul,aten::fill_,aten::zero_,aten::sum,aten::cat,aten::gelu,aten::gelu_backward,aten::div,aten::sub,aten::clone,aten::contiguous,aten::_to_copy,aten::add_,aten::mul_,aten::upsample_bilinear2d,aten::upsample_bilinear2d_backward,aten::sigmoid,aten::elu,aten::elu_backward,aten::threshold_backward,aten::relu" timeout 600 python scripts/probe_step_ops.py > $out/ops_shapes.txt 2>&1
DD_PROBE_STACK=1 DD_PROBE_ROWS=90 DD_PROBE_FILTER="aten::add,aten::copy_,aten::mul,aten::addcmul,aten::fill_,aten::zero_,aten::sum,aten::cat,aten::add_,aten::mul_,aten::div,aten::clone" timeout 600 python scripts/probe_step_ops.py > $out/ops_stack.txt 2>&1
tail -60 $out/ops_stack.txt
