cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dbg
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests -m gpu -x -s -q -p no:cacheprovider > gpurun_out/dbg/run$i.log 2>&1; rc=$?
  echo "run $i rc=$rc"; 
  if [ $rc -ne 0 ]; then grep -v "^$" gpurun_out/dbg/run$i.log | grep -v "File \"/usr" | tail -25; fi
done
