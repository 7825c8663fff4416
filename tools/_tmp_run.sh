for g in 8 16 24 32 64; do
  echo "group $g"
  CDS_SCAN_MASKS_PER_LAUNCH=$g MASKS=64 DBGS="0" bash tools/ablate_multi.sh
done
