// rt_opts_reset.h -- forget every per-instance option before the next instantiation of
// rt_trace.inc.h / rt_sched.inc.h (no include guard: meant to be included repeatedly).
#undef RT_NS
#undef RT_KERNEL_NAME
#undef RT_SCHED_KERNEL_NAME
#undef RT_OPT_UNROLL
#undef RT_OPT_SKIPNEG
#undef RT_OPT_STAMPS
#undef RT_OPT_WAVE_TILE_W
#undef RT_OPT_COOP
#undef RT_OPT_MINWAVES
#undef RT_OPT_LEAN_SQRT
#undef RT_OPT_PERSIST
#undef RT_OPT_LEAN_RCP
#undef RT_OPT_SQRT_NOCHECK
#undef RT_OPT_SHORT_ROOTS
#undef RT_OPT_JOINT_SKIP
#undef RT_OPT_ANY_JOINT
#undef RT_OPT_GLOSS_ID
#undef RT_OPT_TIMELOG
#undef RT_OPT_AB_OLD
#undef RT_PACK_KERNEL_NAME
#undef RT_OPT_WG_WAVES
#undef RT_OPT_BVH
#undef RT_WALK_RAYS_KERNEL_NAME
#undef RT_OPT_GLOBAL_TABLES
