O=$GRAFT_REPO_ROOT/gpurun_out/r5n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/train_profile.py --graph 3 > $O/train.log 2>&1
db=$(find $O/prof -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/trace_report.py $db --sequence 640 > $O/seq.txt
rm -rf $O/prof
grep -n "contract_dma_kernel\|wgrad_kernel\|contract_kernelILi1" $O/seq.txt | tail -60 | cut -c1-140
