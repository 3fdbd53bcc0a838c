for g in T S; do for sf in 0 17 20 25 29 34 50 64; do
  if [ $sf = 0 ]; then r=$(python3 scripts/prof_run.py 256 400 60 $g | tail -1); else r=$(GOMEL_SEG_FRAMES=$sf python3 scripts/prof_run.py 256 400 60 $g | tail -1); fi
  echo "$g seg_frames $sf: $r" | sed "s/'total_ms'.*'dom_ms': \([0-9.]*\).*/dom_ms \1/"
done; done
