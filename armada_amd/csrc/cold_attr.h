// cold_attr.h — bisect hook for DESIGN.md §9 ("minsize on the cold functions gives a different round"): -DCOLD_MINSIZE_MASK=<bits> puts
// __attribute__((minsize)) on the selected out-of-line functions of the generic path (tools/minsize_bisect.sh).  The product build leaves the mask at 0.
#pragma once
#ifndef COLD_MINSIZE_MASK
#define COLD_MINSIZE_MASK 0
#endif
#if (COLD_MINSIZE_MASK >> 0) & 1
#define COLD_MS_0 __attribute__((minsize))
#else
#define COLD_MS_0
#endif
#if (COLD_MINSIZE_MASK >> 1) & 1
#define COLD_MS_1 __attribute__((minsize))
#else
#define COLD_MS_1
#endif
#if (COLD_MINSIZE_MASK >> 2) & 1
#define COLD_MS_2 __attribute__((minsize))
#else
#define COLD_MS_2
#endif
#if (COLD_MINSIZE_MASK >> 3) & 1
#define COLD_MS_3 __attribute__((minsize))
#else
#define COLD_MS_3
#endif
#if (COLD_MINSIZE_MASK >> 4) & 1
#define COLD_MS_4 __attribute__((minsize))
#else
#define COLD_MS_4
#endif
#if (COLD_MINSIZE_MASK >> 5) & 1
#define COLD_MS_5 __attribute__((minsize))
#else
#define COLD_MS_5
#endif
#if (COLD_MINSIZE_MASK >> 6) & 1
#define COLD_MS_6 __attribute__((minsize))
#else
#define COLD_MS_6
#endif
#if (COLD_MINSIZE_MASK >> 7) & 1
#define COLD_MS_7 __attribute__((minsize))
#else
#define COLD_MS_7
#endif
#if (COLD_MINSIZE_MASK >> 8) & 1
#define COLD_MS_8 __attribute__((minsize))
#else
#define COLD_MS_8
#endif
#if (COLD_MINSIZE_MASK >> 9) & 1
#define COLD_MS_9 __attribute__((minsize))
#else
#define COLD_MS_9
#endif
#if (COLD_MINSIZE_MASK >> 10) & 1
#define COLD_MS_10 __attribute__((minsize))
#else
#define COLD_MS_10
#endif
#if (COLD_MINSIZE_MASK >> 11) & 1
#define COLD_MS_11 __attribute__((minsize))
#else
#define COLD_MS_11
#endif
#if (COLD_MINSIZE_MASK >> 12) & 1
#define COLD_MS_12 __attribute__((minsize))
#else
#define COLD_MS_12
#endif
#if (COLD_MINSIZE_MASK >> 13) & 1
#define COLD_MS_13 __attribute__((minsize))
#else
#define COLD_MS_13
#endif
#if (COLD_MINSIZE_MASK >> 14) & 1
#define COLD_MS_14 __attribute__((minsize))
#else
#define COLD_MS_14
#endif
#if (COLD_MINSIZE_MASK >> 15) & 1
#define COLD_MS_15 __attribute__((minsize))
#else
#define COLD_MS_15
#endif
