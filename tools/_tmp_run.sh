for g in 32 48 64 96 128; do
  echo "group $g"
  CDS_SCAN_MASKS_PER_LAUNCH=$g MASKS=128 TARGETS=8192 DBGS="0" bash tools/ablate_multi.sh
done
