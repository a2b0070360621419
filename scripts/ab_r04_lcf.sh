#!/bin/bash
# frames per band-level launch inside a 1024-frame pass (TACEX_LEVEL_CHUNK_FRAMES), one box, two rounds
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04lcf}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lcf in 256 171 205 342 512 128; do
    TACEX_LEVEL_CHUNK_FRAMES=$lcf timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']; print('lcf $lcf rep $rep: C3', d['value'], d['ms_per_step'], {k: (round(v['avg_ms']*1e3,1), v['launches_per_update']) for k, v in st.items()}, 'levels per update [us]', round(sum(v['avg_ms']*v['launches_per_update'] for k,v in st.items() if k.startswith('blur'))*1e3,1))" | tee -a $OUT/ab.log
  done
done
