# GPU box: scripts/exp_poison.py under three fill patterns, results compared bit for bit.  usage: bash scripts/exp_poison.sh [preset]
cd $GRAFT_REPO_ROOT
P=${1:-B}
for pat in 0x00 0xFF 0x4B; do timeout -k 10 300 python scripts/exp_poison.py $pat $P /tmp/poison_$pat.pt 2>&1 | grep pattern || exit 1; done
python - <<'PY'
import torch
a = torch.load('/tmp/poison_0x00.pt')
for pat in ('0xFF', '0x4B'):
    b = torch.load(f'/tmp/poison_{pat}.pt')
    print(f'{pat} against 0x00: max |d w| {float((a["w"] - b["w"]).abs().max()):.3e}, max |d img| {float((a["img"] - b["img"]).abs().max()):.3e}')
PY
