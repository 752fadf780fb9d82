// sl_gp4_clobbers.h - accumulator registers of k_gp_sweep4 by (row block r, cell-block pair h):
// acc(r, cb, rot) = a[2 i : 2 i + 1], i = (r * 4 + cb) * 4 + rot, so cell blocks 2h, 2h + 1 of row block r
// live in a[32 r + 16 h : 32 r + 16 h + 15].  Clobber lists of inline asm must be string literals,
// hence this table (generated: tools/gen_gp4_clobbers.py).
#pragma once
#define SL_GP4_CL_0_0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define SL_GP4_CL_0_1 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
#define SL_GP4_CL_0 SL_GP4_CL_0_0, SL_GP4_CL_0_1
#define SL_GP4_CL_1_0 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47"
#define SL_GP4_CL_1_1 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
#define SL_GP4_CL_1 SL_GP4_CL_1_0, SL_GP4_CL_1_1
#define SL_GP4_CL_2_0 "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79"
#define SL_GP4_CL_2_1 "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
#define SL_GP4_CL_2 SL_GP4_CL_2_0, SL_GP4_CL_2_1
#define SL_GP4_CL_3_0 "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111"
#define SL_GP4_CL_3_1 "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define SL_GP4_CL_3 SL_GP4_CL_3_0, SL_GP4_CL_3_1
#define SL_GP4_CL_4_0 "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143"
#define SL_GP4_CL_4_1 "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159"
#define SL_GP4_CL_4 SL_GP4_CL_4_0, SL_GP4_CL_4_1
#define SL_GP4_CL_5_0 "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175"
#define SL_GP4_CL_5_1 "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"
#define SL_GP4_CL_5 SL_GP4_CL_5_0, SL_GP4_CL_5_1
#define SL_GP4_CL_6_0 "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207"
#define SL_GP4_CL_6_1 "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223"
#define SL_GP4_CL_6 SL_GP4_CL_6_0, SL_GP4_CL_6_1
#define SL_GP4_CL_7_0 "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239"
#define SL_GP4_CL_7_1 "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define SL_GP4_CL_7 SL_GP4_CL_7_0, SL_GP4_CL_7_1
