# dev tool: the two-stream non-reproducibility under a list of configurations (scripts/exp_overlap_bisect.py)
cd $GRAFT_REPO_ROOT
P="python scripts/exp_overlap_bisect.py"
$P --tag base
$P --tag eager --no-graph
$P --tag f32 --precision f32 --steps 4
$P --tag bf16x3 --precision bf16x3 --steps 8
LA_NO_XS_HANDOFF=1 $P --tag no_xs_handoff
LA_NO_SEAM_FUSE=1 LA_NO_SEAM2_FUSE=1 $P --tag no_seam_fuse
$P --tag fwd_only --fwd-only --steps 20
$P --tag no_pix --w-pix 0
$P --tag res128 --res 128
