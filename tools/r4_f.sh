cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "" "noopt" "noattn2"; do
echo "=== variant: $v"
timeout 300 python tools/r4/nan_hunt.py $v 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|Consider\|print(" | grep -E "^[0-9] (src|img|logits|non-finite grads|loss)" | cut -c1-260
done
