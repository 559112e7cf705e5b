# the round's whole profile pass in one call: every workload with the schedule its tuner keeps, then the runner-up schedules
bash scripts/profile_round.sh gpurun_out/r06q cornell cornell_native sponza sponza_walk8 sponza_walk8_native sponza_walk8c dragon dragon_walk8 dragon_walk8_native dragon_walk8c hairball hairball_4k hairball_4k_walk8 hairball_4k_walk8_native hairball_4k_walk8c > gpurun_out/r06q.log 2>&1
bash scripts/profile_round.sh gpurun_out/r06q sponza_p4 sponza_p6 sponza_walk8_p4 sponza_walk8_p6 sponza_walk8_native_p4 sponza_walk8_native_p6 sponza_walk8c_p4 sponza_walk8c_p6 dragon_walk8c_p4 dragon_walk8c_p6 hairball_4k_walk8c_p4 hairball_4k_walk8c_p6 hairball_4k_walk8c_p2 >> gpurun_out/r06q.log 2>&1
tail -30 gpurun_out/r06q.log
