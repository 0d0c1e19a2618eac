#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export BTX_NO_TALL=1
for sh in 64,64,56 512,512,7; do
python tools/libcmp.py run /tmp/ref.pt $sh 2>&1 | grep -v amdgpu.ids
BTX_LIB=$PWD/build_variants/libbtx_tune.so python tools/libcmp.py run /tmp/new.pt $sh 2>&1 | grep -v amdgpu.ids
BTX_PERSIST=1 BTX_LIB=$PWD/build_variants/libbtx_tune.so python tools/libcmp.py run /tmp/per.pt $sh 2>&1 | grep -v amdgpu.ids
echo "committed lib vs sectioned taps kernel:"; python tools/libcmp.py cmp /tmp/new.pt /tmp/ref.pt
echo "committed lib vs persistent:"; python tools/libcmp.py cmp /tmp/per.pt /tmp/ref.pt
python tools/libcmp.py run /tmp/ref.pt $sh --res 2>&1 | grep -v amdgpu.ids
BTX_PERSIST=1 BTX_LIB=$PWD/build_variants/libbtx_tune.so python tools/libcmp.py run /tmp/per.pt $sh --res 2>&1 | grep -v amdgpu.ids
echo "fused (bn+res+relu): committed lib vs persistent:"; python tools/libcmp.py cmp /tmp/per.pt /tmp/ref.pt
done
