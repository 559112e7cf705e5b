# thresholds of the lane state machine (phased-mid): lanes that leave a node phase before it ends x lanes that wait before a shade phase runs
export PBR_PLAN=4
for pk in 8 16 24 32; do for sh in 24 32 40 48; do
  echo "== PH_PARK $pk PH_SHADE $sh"; PBR_PH_PARK=$pk PBR_PH_SHADE=$sh bash scripts/lab_run.sh "sponza:32 dragon:32 hairball:16" cur 2>&1 | grep Msamples | cut -c48-100
done; done
