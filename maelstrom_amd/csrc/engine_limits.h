// engine_limits.h — compile-time limits shared by the host and device parts of libmaelsim.
#ifndef MSIM_ENGINE_LIMITS_H
#define MSIM_ENGINE_LIMITS_H
#define MSIM_MAX_NODES 128u
#define MSIM_MASK_WORDS 4u      /* node-set masks are MSIM_MAX_NODES/32 words (grudge rows in the payload) */
#endif
