for pad in 0 4 20; do
  echo "ringpad $pad"
  CDS_SCAN_RINGPAD=$pad DBGS="0" bash tools/ablate.sh
  CDS_SCAN_RINGPAD=$pad DBGS="0" bash tools/ablate_multi.sh
done
