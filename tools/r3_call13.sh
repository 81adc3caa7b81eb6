cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gimm-vfi_amd/lib
for i in 1 2; do for v in A B C; do
  if [ $v = C ]; then unset GVFI_LIB_PATH; else export GVFI_LIB_PATH=$L/libgimmvfi_hip_$v.so; fi
  echo "R $v: $(timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done; done
for v in A B C; do
  if [ $v = C ]; then unset GVFI_LIB_PATH; else export GVFI_LIB_PATH=$L/libgimmvfi_hip_$v.so; fi
  echo "F $v: $(timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done
