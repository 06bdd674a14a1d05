R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; cp evogp_amd/lib/libevogp_hip.so /tmp/keep.so
for lib in default nostage stageonly; do
  [ $lib = default ] && cp /tmp/keep.so evogp_amd/lib/libevogp_hip.so || cp evogp_amd/lib/libevogp_hip_$lib.so evogp_amd/lib/libevogp_hip.so
  echo "#### $lib"; SIZES="64 4096 100000" bash scripts/dbg/launch_floor.sh x 2>&1 | grep "trees:\|sr_tc_kernel" 
done
cp /tmp/keep.so evogp_amd/lib/libevogp_hip.so
