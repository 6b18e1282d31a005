#!/bin/bash
# Which of the kept-negative patches under scripts/exp/ still apply?  Each is a `git diff` against the commit it was measured on (the
# table in scripts/README.md); they are history, not maintained against HEAD -- this reports, per patch, whether `git apply --check`
# succeeds against its base commit (always expected) and against HEAD (informative: what it would take to re-run the experiment).
#   usage: bash scripts/exp/check_patches.sh        (from the repository root; touches nothing: uses a temporary worktree per base)
cd "$(git rev-parse --show-toplevel)" || exit 1
declare -A BASE=(
  [lin_halo_kernel.patch]=68a3dd7 [k4_kernel.patch]=fb43017 [halo_mt64_m16.patch]=5eba378 [splitk_fused_reduce.patch]=aea95d9
  [halo_splitk_small_grids.patch]=884a882 [image_chain_side_stream.patch]=4efbee0 [splitk_direct_fp32_loader.patch]=a6f3349
  [disc_skip_branch_side_stream.patch]=f51b360 [halo_persistent_small_tiles.patch]=198c70d
  [epilogue_wide_stores.patch]=bd209d2 [epilogue_straightline_and_stagger.patch]=f8bfd28
)
rc=0
for p in scripts/exp/*.patch; do
  n=$(basename "$p"); b=${BASE[$n]}
  head_ok=no; git apply --check "$p" 2>/dev/null && head_ok=yes
  base_ok="?"
  if [ -n "$b" ] && git cat-file -e "$b^{commit}" 2>/dev/null; then
    t=$(mktemp -d); git worktree add -q --detach "$t" "$b" 2>/dev/null && { (cd "$t" && git apply --check "$OLDPWD/$p" 2>/dev/null) && base_ok=yes || base_ok=no; git worktree remove --force "$t"; }
  fi
  printf '%-42s base %-8s applies to base: %-3s to HEAD: %s\n' "$n" "${b:-unknown}" "$base_ok" "$head_ok"
  [ "$base_ok" = no ] && rc=1
done
exit $rc
