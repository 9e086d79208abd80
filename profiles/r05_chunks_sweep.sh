for c in cfg2 cfg3; do for ch in 4 8 16; do bash profiles/kstat.sh $c "KFH_INTEGRATE_CHUNKS=$ch" | head -1 | sed "s/^/$c /"; done; done
