for t in "" "wbranch=1" "klz2_main=0" "klz2_main=1" "fold_join=0" "tail_gate=0" "fprop_tail=0" "raw_heads=0"; do
echo "TUNE=$t: $(DRVAE_TUNE=$t timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['steady_state']['ms_per_step'], d.get('chain_wait_us'), d['config'].get('side_cus'))")"
done
