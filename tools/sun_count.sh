cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for f in "-DINCR_DIAG_SUN_COUNT=2"; do
rm -f raster.o raster_api.o; make -s -j16 EXTRA="$f" > /dev/null 2>&1
(cd $GRAFT_REPO_ROOT && python tools/raster_time.py --texture-size 256 --frames 1 2>&1 | grep -E "shadow rays|sun cover" | sort | uniq -c | sort -k3,3n -k4,4n)
done
rm -f raster.o raster_api.o; make -s -j16 > /dev/null 2>&1
