bash scripts/profile_round.sh gpurun_out/r06p sponza_p4 sponza_walk8_p4 sponza_walk8_native_p4 sponza_walk8c_p6 dragon_walk8c_p6 hairball_4k_walk8c_p4 > gpurun_out/r06p_alt.log 2>&1
tail -8 gpurun_out/r06p_alt.log
