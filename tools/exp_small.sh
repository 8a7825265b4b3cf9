cd "${GRAFT_REPO_ROOT:-/root/repo}"
for S in 4 32 46 64 96; do
  for p in "" "split=1" "split=2" "split=4"; do
    t=$(timeout -k 10 100 python tools/yeast_time.py "$p" $S 2>&1 | tail -1 | sed 's/.*K1 \([0-9.]*\) ms.*/\1/')
    echo "yeast S=$S pairs=$((S*(S-1)/2)) plan='$p' K1=$t"
  done
done
