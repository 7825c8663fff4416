for g in 128 256; do
  echo "group $g"
  CDS_SCAN_MASKS_PER_LAUNCH=$g MASKS=256 TARGETS=16384 DBGS="0" bash tools/ablate_multi.sh
done
