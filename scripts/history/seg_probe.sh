# alternating A/B of the fused kernel's segment plan at full batches (same box, same call): the planner's choice against forced lengths
for g in T D; do for i in 1 2 3; do
  for sf in 0 100 67; do
    if [ $sf = 0 ]; then r=$(python3 scripts/prof_run.py 256 400 60 $g | tail -1); else r=$(GOMEL_SEG_FRAMES=$sf python3 scripts/prof_run.py 256 400 60 $g | tail -1); fi
    echo "$g seg_frames $sf: $(echo "$r" | sed "s/.*'dom_ms': \([0-9.]*\).*/\1/") ms / 60 launches"
  done
done; done
