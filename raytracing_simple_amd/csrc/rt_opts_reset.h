// rt_opts_reset.h -- forget every per-instance option before the next instantiation of
// rt_trace.inc.h / rt_sched.inc.h (no include guard: meant to be included repeatedly).
#undef RT_NS
#undef RT_KERNEL_NAME
#undef RT_SCHED_KERNEL_NAME
#undef RT_PACK_KERNEL_NAME
#undef RT_WALK_RAYS_KERNEL_NAME
#undef RT_OPT_WG_WAVES
#undef RT_OPT_COOP
#undef RT_OPT_WALK
#undef RT_OPT_GLOBAL_TABLES
#undef RT_OPT_MINWAVES
#undef RT_OPT_PERSIST
#undef RT_OPT_STAMPS
#undef RT_OPT_TIMELOG
#undef RT_OPT_EXACT_DECISIONS
#undef RT_OPT_PAIR_PLANES
#undef RT_OPT_RAYS2

#undef RT_NO_RENDER_KERNEL
#undef RT_OPT_TOP_PAIRS
#undef RT_OPT_PREFETCH
#undef RT_OPT_PACKED_PAIRS
