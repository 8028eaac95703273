// The one-block-per-CU shape of the fused edge transition (csrc/fd_edge_mlp.hip has the kernel; this translation unit instantiates
// it with 8 waves x 16 rows = 128-row tiles, 48 KB weight stages of four units, ring of two = 96 KB of LDS, <= 256 registers).
// Against the two-blocks-per-CU shape: ONE weight stream per CU (half the L2 -> LDS traffic per pair row) and a stage barrier every
// 96 instead of every 48 MFMAs of a wave; fd_edge_mlp() picks it from FD_EDGE_MLP_W8_MIN_ROWS pair rows up.
#define EM_SHAPE_W8
#define EM_WAVES 8
#define EM_UPS 4
#include "fd_edge_mlp.hip"
