"""ctypes binding of libgkrmsm_hip.so (the C ABI in include/gkrmsm.h).

This is harness plumbing for tests and bench.py -- the product is the shared library.  There is no
fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GM_LIB_PATH") or os.path.join(_HERE, "libgkrmsm_hip.so")   # GM_LIB_PATH: an alternative build (A/B measurements)


class GmError(RuntimeError):
    pass


class GmFn(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("prim", C.c_int32 * 4), ("count", C.c_int32 * 4)]


FN_AFF_L1, FN_AFF_L2, FN_AFF_L3, FN_PROJ_L1, FN_PROJ_L2, FN_PROJ_L3 = 1, 2, 3, 4, 5, 6
FN_TRI_L1, FN_ID, FN_BITCHECK, FN_PT_BIT_CHOICE, FN_ADD_INVERSES, FN_LOGUP_LAYER = 7, 8, 9, 10, 11, 12


def make_fn(*segs):
    """make_fn((prim, count), ...)  ->  GmFn  (Stacked / Repeated composition, left to right)"""
    f = GmFn()
    f.nseg = len(segs)
    for i, (p, c) in enumerate(segs):
        f.prim[i] = p
        f.count[i] = c
    return f


_lib = None

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16)
vp = C.c_void_p

WRITE_SCALARS_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64)
CHALLENGE_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64))


class GmTranscript(C.Structure):
    """gm_transcript: the caller's live Fiat-Shamir transcript as two callbacks"""
    _fields_ = [("ctx", C.c_void_p), ("write_scalars", WRITE_SCALARS_CB), ("challenge", CHALLENGE_CB),
                ("write_points", WRITE_SCALARS_CB)]


READ_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64))


class GmTranscriptReader(C.Structure):
    """gm_transcript_reader: the caller's transcript in verifier mode"""
    _fields_ = [("ctx", C.c_void_p), ("read_scalars", READ_CB), ("challenge", CHALLENGE_CB), ("read_points", READ_CB),
                ("points_validated", C.c_uint32), ("reserved", C.c_uint32)]


ALL_GATHER_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64)


class GmFragment(C.Structure):
    """gm_fragment: Fragment{mem_idx, len, content, start} of a gen-1 Shape (content 0 = Data, 1 = Consts)"""
    _fields_ = [("mem_idx", C.c_uint64), ("len", C.c_uint64), ("start", C.c_uint64), ("content", C.c_uint32),
                ("reserved", C.c_uint32)]


class GmScProfileRow(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("launches", C.c_uint32), ("k_cols", C.c_uint32), ("total_ms", C.c_double),
                ("max_ms", C.c_double), ("pairs", C.c_double), ("alg_bytes", C.c_double), ("fr_mul", C.c_double),
                ("max_ms_pairs", C.c_double)]


ALL_GATHER_DEV_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class GmKeyView(C.Structure):
    _fields_ = [("n_segments", C.c_uint32), ("reserved", C.c_uint32), ("d_segment", C.POINTER(C.c_void_p)),
                ("first", C.POINTER(C.c_uint64)), ("count", C.POINTER(C.c_uint64))]


class GmComm(C.Structure):
    """gm_comm: rank / world, one host-buffer all-gather and, optionally (NULL by default), the same collective on device buffers"""
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("all_gather", ALL_GATHER_CB),
                ("all_gather_dev", ALL_GATHER_DEV_CB), ("pull_dev", C.c_void_p)]


_SIGS = {
    "gm_last_error": (C.c_char_p, []),
    "gm_version": (C.c_char_p, []),
    "gm_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "gm_set_device": (C.c_int32, [C.c_int32]),
    "gm_stream_sync": (C.c_int32, [vp]),
    "gm_stream_create": (C.c_int32, [C.POINTER(vp)]),
    "gm_stream_destroy": (C.c_int32, [vp]),
    "gm_malloc": (C.c_int32, [C.POINTER(vp), C.c_size_t]),
    "gm_free": (C.c_int32, [vp]),
    "gm_release_cached_memory": (C.c_int32, []),
    "gm_reserve": (C.c_int32, [C.c_uint64]),
    "gm_unreserve": (C.c_int32, []),
    "gm_memory_stats": (C.c_int32, [u64p]),
    "gm_set_wait_timeout_ms": (C.c_int32, [C.c_uint32]),
    "gm_stage_slots": (C.c_int32, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gm_sc_profile": (C.c_int32, [C.c_int32]),
    "gm_sc_profile_read": (C.c_int32, [C.POINTER(GmScProfileRow), C.c_uint32, u32p, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]),
    "gm_memcpy_h2d": (C.c_int32, [vp, vp, C.c_size_t, vp]),
    "gm_memcpy_d2h": (C.c_int32, [vp, vp, C.c_size_t, vp]),
    "gm_memcpy_d2d": (C.c_int32, [vp, vp, C.c_size_t, vp]),
    "gm_fn_shape": (C.c_int32, [C.POINTER(GmFn), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gm_fr_batch": (C.c_int32, [C.c_int32, vp, vp, vp, C.c_uint64, vp]),
    "gm_fr_host": (C.c_int32, [C.c_int32, vp, vp, vp, C.c_uint64]),
    "gm_fn_host": (C.c_int32, [C.POINTER(GmFn), vp, vp, C.c_uint64]),
    "gm_dense_map": (C.c_int32, [C.POINTER(GmFn), vp, vp, C.c_uint64, vp]),
    "gm_dense_map_split": (C.c_int32, [C.POINTER(GmFn), vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp]),
    "gm_dense_bind": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint64, vp, vp]),
    "gm_eq_table": (C.c_int32, [vp, vp, C.c_uint32, vp, vp, vp]),
    "gm_vv_from_host": (C.c_int32, [C.c_uint32, C.c_uint32, vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp), vp]),
    "gm_vv_from_msm": (C.c_int32, [vp, vp, C.c_uint32, C.POINTER(vp), vp]),
    "gm_vv_map": (C.c_int32, [C.POINTER(GmFn), vp, C.POINTER(vp), vp]),
    "gm_vv_map_split": (C.c_int32, [C.POINTER(GmFn), vp, C.c_uint32, C.POINTER(vp), vp]),
    "gm_vv_map_split_to_dense": (C.c_int32, [C.POINTER(GmFn), vp, C.c_uint32, vp, vp]),
    "gm_vv_slice": (C.c_int32, [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "gm_vv_concat": (C.c_int32, [vp, vp, C.POINTER(vp)]),
    "gm_vv_to_dense": (C.c_int32, [vp, vp, vp]),
    "gm_vv_info": (C.c_int32, [vp, u32p, u32p, u64p, u32p, u32p]),
    "gm_vv_read": (C.c_int32, [vp, C.c_uint32, vp, vp, vp]),
    "gm_vv_pads": (C.c_int32, [vp, vp, vp]),
    "gm_vv_destroy": (C.c_int32, [vp]),
    "gm_sc_dense_deg2_create": (C.c_int32, [C.POINTER(GmFn), C.c_uint32, vp, vp, vp, vp, C.POINTER(vp), vp]),
    "gm_sc_vecvec_deg2_create": (C.c_int32, [C.POINTER(GmFn), vp, vp, vp, vp, C.POINTER(vp), vp]),
    "gm_sc_dense_create": (C.c_int32, [C.c_int32, C.POINTER(GmFn), C.c_uint32, vp, vp, vp, C.POINTER(vp), vp]),
    "gm_sc_unipoly": (C.c_int32, [vp, vp, u32p]),
    "gm_sc_bind": (C.c_int32, [vp, vp]),
    "gm_sc_final_evals": (C.c_int32, [vp, vp, u32p]),
    "gm_sc_claim": (C.c_int32, [vp, vp]),
    "gm_sc_destroy": (C.c_int32, [vp]),
    "gm_pip_witness_create": (C.c_int32, [vp, vp, C.c_uint32, C.POINTER(vp), vp]),
    "gm_pip_witness_create_sharded": (C.c_int32, [vp, vp, C.c_uint32, C.POINTER(GmComm), C.POINTER(vp), vp]),
    "gm_comm_sum_fr": (C.c_int32, [C.POINTER(GmComm), vp, C.c_uint32]),
    "gm_shard_clock": (C.c_int32, [C.c_int32, C.POINTER(C.c_double)]),
    "gm_msm_g1_outer_part": (C.c_int32, [vp, vp, vp, C.c_uint32, vp, vp, C.c_uint64, u32p, u32p, u32p, vp, vp, vp]),
    "gm_g1_combine_parts": (C.c_int32, [C.POINTER(GmComm), vp, C.c_uint32, vp]),
    "gm_comm_rccl_unique_id": (C.c_int32, [vp]),
    "gm_comm_rccl_create": (C.c_int32, [vp, C.c_uint32, C.c_uint32, C.POINTER(vp), vp]),
    "gm_comm_rccl_destroy": (C.c_int32, [vp]),
    "gm_comm_rccl_as_comm": (C.c_int32, [vp, C.POINTER(GmComm)]),
    "gm_comm_rccl_all_gather_dev": (C.c_int32, [vp, vp, vp, C.c_uint64, vp]),
    "gm_comm_rccl_broadcast_dev": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, vp]),
    "gm_comm_rccl_stats": (C.c_int32, [vp, u64p, u64p]),
    "gm_sc_stage_counts": (C.c_int32, [u64p, u64p]),
    "gm_comm_shm_create": (C.c_int32, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "gm_comm_shm_destroy": (C.c_int32, [vp]),
    "gm_comm_shm_as_comm": (C.c_int32, [vp, C.POINTER(GmComm)]),
    "gm_comm_shm_stats": (C.c_int32, [vp, u64p, u64p]),
    "gm_comm_shm_ipc_stats": (C.c_int32, [vp, u64p, u64p, u64p]),
    "gm_pip_witness_destroy": (C.c_int32, [vp]),
    "gm_pip_witness_outputs": (C.c_int32, [vp, vp, u32p, u64p, vp]),
    "gm_pip_witness_bytes": (C.c_uint64, [vp]),
    "gm_pip_prove_image_part": (C.c_int32, [vp, vp, vp, vp, C.c_uint64, vp, C.c_uint64, u64p, vp, u32p, vp, u64p, u64p]),
    "gm_pip_prove_image_part_tr": (C.c_int32, [vp, vp, vp, C.POINTER(GmTranscript), vp, u32p, vp, u64p, u64p]),
    "gm_gkr_msm_prove_tr": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(GmTranscript), vp, vp, u32p, vp, u64p, u64p,
                                        vp]),
    "gm_pushforward_prove": (C.c_int32, [vp, vp, C.c_uint32, vp, vp, vp, C.c_uint64, vp, C.c_uint64, u64p, vp, vp, vp, vp, vp, vp, vp,
                                         u64p, u64p, vp]),
    "gm_pushforward_prove_sharded": (C.c_int32, [vp, vp, C.c_uint32, C.POINTER(GmComm), vp, vp, vp, C.c_uint64, vp, C.c_uint64, u64p, vp,
                                                 vp, vp, vp, vp, vp, vp, u64p, u64p, vp]),
    "gm_pushforward_prove_tr": (C.c_int32, [vp, vp, C.c_uint32, vp, vp, C.POINTER(GmTranscript), vp, vp, vp, vp, vp, vp, vp, u64p,
                                            u64p, vp]),
    "gm_multiopen_prove": (C.c_int32, [C.c_uint32, C.c_uint32, vp, vp, vp, vp, C.c_uint64, vp, C.c_uint64, u64p, vp, vp, u64p, u64p, vp]),
    "gm_multiopen_prove_tr": (C.c_int32, [C.c_uint32, C.c_uint32, vp, vp, vp, C.POINTER(GmTranscript), vp, vp, u64p, u64p, vp]),
    "gm_pippenger_last_spans": (C.c_int32, [C.POINTER(C.c_double)]),
    "gm_pippenger_wg_create": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, vp, C.POINTER(vp), vp]),
    "gm_pippenger_wg_destroy": (C.c_int32, [vp]),
    "gm_pippenger_wg_witness": (C.c_int32, [vp, C.POINTER(vp)]),
    "gm_pippenger_prove": (C.c_int32, [vp, vp, vp, vp, vp, vp, C.c_uint64, vp, C.c_uint64, u64p, vp, C.c_uint64, u64p, vp, u64p, u64p]),
    "gm_pippenger_prove_tr": (C.c_int32, [vp, vp, vp, vp, vp, C.POINTER(GmTranscript), vp, u64p, u64p]),
    "gm_merlin_create": (C.c_int32, [vp, C.c_uint64, C.POINTER(vp)]),
    "gm_merlin_destroy": (C.c_int32, [vp]),
    "gm_merlin_transcript": (C.c_int32, [vp, C.POINTER(GmTranscript)]),
    "gm_merlin_proof": (C.c_int32, [vp, C.POINTER(vp), u64p]),
    "gm_merlin_append_message": (C.c_int32, [vp, vp, C.c_uint64, vp, C.c_uint64]),
    "gm_merlin_challenge_bytes": (C.c_int32, [vp, vp, C.c_uint64, vp, C.c_uint64]),
    "gm_keccak_f1600": (C.c_int32, [vp]),
    "gm_pippenger_verify_tr": (C.c_int32, [C.c_uint32] * 5 + [vp, vp, vp, vp, C.POINTER(GmTranscriptReader), vp]),
    "gm_pippenger_verify": (C.c_int32, [C.c_uint32] * 5 + [vp, vp, vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, vp, u64p]),
    "gm_kzg_verify_pair": (C.c_int32, [vp, vp, vp]),
    "gm_gkr_msm_verify": (C.c_int32, [C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, C.c_uint64, vp, u32p, vp, u64p, u64p]),
    "gm_gkr_msm_verify_tr": (C.c_int32, [C.c_uint32, C.c_uint32, C.POINTER(GmTranscriptReader), vp, u32p, vp, u64p]),
    "gm_pairing": (C.c_int32, [vp, vp, vp]),
    "gm_kzg_mock_vk": (C.c_int32, [vp, vp, vp]),
    "gm_merlin_create_verifier": (C.c_int32, [vp, C.c_uint64, vp, C.c_uint64, C.POINTER(vp)]),
    "gm_merlin_reader": (C.c_int32, [vp, C.POINTER(GmTranscriptReader)]),
    "gm_merlin_unread": (C.c_int32, [vp, u64p]),
    "gm_kzg_div_by_linear": (C.c_int32, [vp, C.c_uint64, vp, vp, vp, vp]),
    "gm_knuckles_setup": (C.c_int32, [vp, C.c_uint32, vp, vp]),
    "gm_knuckles_setup_range": (C.c_int32, [vp, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp]),
    "gm_knuckles_open_sharded": (C.c_int32, [C.POINTER(GmComm), C.POINTER(GmKeyView), vp, vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint64,
                                             vp, vp, vp]),
    "gm_knuckles_open_sharded_tr": (C.c_int32, [C.POINTER(GmComm), C.POINTER(GmKeyView), vp, vp, C.c_uint32, vp, vp, vp, vp,
                                                C.POINTER(GmTranscript), vp, vp, vp]),
    "gm_pippenger_wg_create_sharded": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(GmKeyView), C.POINTER(GmComm), C.POINTER(vp), vp]),
    "gm_pippenger_sharded_key_ranges": (C.c_int32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u64p, u64p]),
    "gm_knuckles_open": (C.c_int32, [vp, vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, vp, vp, C.c_uint64, vp, vp, vp]),
    "gm_knuckles_open_tr": (C.c_int32, [vp, vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, vp, C.POINTER(GmTranscript), vp, vp, vp]),
    "gm_gkr_msm_prove": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, C.c_uint64, u64p, vp, vp, u32p, vp,
                                     u64p, u64p, C.POINTER(C.c_double), vp]),
    "gm_msm_plan_create": (C.c_int32, [C.c_uint32] * 5 + [C.POINTER(vp)]),
    "gm_msm_plan_destroy": (C.c_int32, [vp]),
    "gm_msm_plan_workspace_bytes": (C.c_size_t, [vp]),
    "gm_msm_run": (C.c_int32, [vp, vp, vp, vp]),
    "gm_msm_bucket_sums": (C.c_int32, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]),
    "gm_msm_window_points": (C.c_int32, [vp, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gm_msm_digits": (C.c_int32, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    "gm_msm_phase1_polys": (C.c_int32, [vp, vp, vp, vp, vp, vp]),
    "gm_msm_second_phase": (C.c_int32, [vp, vp, C.c_uint32, vp, vp, vp]),
    "gm_msm_profile": (C.c_int32, [vp, C.c_int32]),
    "gm_msm_run_info": (C.c_int32, [vp, C.POINTER(C.c_int32)]),
    "gm_msm_profile_read": (C.c_int32, [vp, C.POINTER(C.c_float), C.c_int32]),
    "gm_msm_level_cells": (C.c_int32, [vp, u64p, C.c_uint32, vp]),
    "gm_msm_combine_host": (C.c_int32, [vp, C.c_uint32, C.c_uint32, vp]),
    "gm_msm_te": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]),
    "gm_g1_msm": (C.c_int32, [vp, vp, C.c_uint64, C.c_int32, C.c_uint32, vp, vp]),
    "gm_g1_msm_nonaff": (C.c_int32, [vp, vp, C.c_uint64, C.c_int32, C.c_uint32, vp, vp]),
    "gm_gkr_msm_commit": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32, vp, vp, vp]),
    "gm_g1_msm_nonaff_grouped": (C.c_int32, [vp, C.c_uint64, vp, C.c_uint32, vp, C.c_int32, C.c_uint32, vp, vp]),
    "gm_g1_bucket_sums": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "gm_g1_pullback_msm": (C.c_int32, [vp, vp, C.c_uint64, vp, C.c_uint32, vp, vp]),
    "gm_g1_weighted_sum": (C.c_int32, [vp, C.c_uint32, C.c_uint32, vp, vp]),
    "gm_g1_prepare_bases": (C.c_int32, [vp, C.c_uint64, C.c_uint32, vp, vp]),
    "gm_g1_binary_msm": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "gm_msm_g1_outer": (C.c_int32, [vp, vp, C.c_uint32, vp, vp, C.c_uint64, u32p, vp, vp, vp]),
    "gm_g1_to_affine": (C.c_int32, [vp, C.c_uint64, vp, vp]),
    "gm_g1_from_affine": (C.c_int32, [vp, C.c_uint64, vp, vp]),
    "gm_g1_host": (C.c_int32, [C.c_int32, vp, vp, vp, C.c_uint64]),
    "gm_g1_batch": (C.c_int32, [C.c_int32, vp, vp, vp, C.c_uint64, vp]),
    "gm_g1_mock_srs": (C.c_int32, [vp, vp, C.c_uint64, vp, vp]),
    "gm_g1_gen_points": (C.c_int32, [vp, C.c_uint64, C.c_uint64, vp]),
    "gm_frag_shape_full_split": (C.c_int32, [C.POINTER(GmFragment), C.c_uint32, C.c_uint64, C.POINTER(GmFragment), C.c_uint32, u32p,
                                             u64p, C.c_uint32, u32p, u64p]),
    "gm_frag_split": (C.c_int32, [C.POINTER(GmFragment), C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]),
    "gm_frag_bind": (C.c_int32, [C.POINTER(GmFragment), C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp]),
    "gm_frag_to_dense": (C.c_int32, [C.POINTER(GmFragment), C.c_uint32, C.c_uint64, vp, vp, vp, vp]),
    "gm_segment_split": (C.c_int32, [C.c_uint64, C.c_uint64, u64p, C.POINTER(C.c_uint8), C.c_uint32, u32p]),
    "gm_frag_eq_materialize": (C.c_int32, [C.POINTER(GmFragment), C.c_uint32, C.c_uint64, vp, vp, C.c_uint32, vp, vp, vp]),
    "gm_triangle_witness_create": (C.c_int32, [vp, C.c_uint32, C.c_uint32, C.POINTER(vp), vp]),
    "gm_bintree_witness_create": (C.c_int32, [vp, C.c_uint32, C.c_int32, C.POINTER(vp), vp]),
    "gm_gkr_witness_destroy": (C.c_int32, [vp]),
    "gm_gkr_witness_output": (C.c_int32, [vp, vp, C.c_uint32, u32p, u32p]),
    "gm_gkr_prove": (C.c_int32, [vp, vp, vp, vp, C.c_uint64, vp, C.c_uint64, u64p, vp, u32p, vp, u32p, u64p, u64p]),
    "gm_gkr_prove_tr": (C.c_int32, [vp, vp, vp, C.POINTER(GmTranscript), vp, u32p, vp, u32p, u64p, u64p]),
    "gm_g1_release_scratch": (C.c_int32, []),
    "gm_g1_generator": (C.c_int32, [vp]),
    "gm_g1_fixed_base_register": (C.c_int32, [vp, C.c_uint64, vp]),
    "gm_g1_fixed_base_release": (C.c_int32, [vp]),
    "gm_pip_witness_claims": (C.c_int32, [vp, vp, vp, u32p]),
    "gm_bs_scalars_into_bigint": (C.c_int32, [vp, vp, C.c_uint64, vp]),
    "gm_gen_points": (C.c_int32, [vp, C.c_uint64, C.c_uint64, vp]),
}


def declared_symbols():
    return sorted(_SIGS)


def lib():
    """Load the shared library (once).  Raises GmError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GmError("libgkrmsm_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(expected at %s)" % LIB_PATH)
        try:  # load torch's HIP runtime first: a second libamdhip64 loaded before it would not see the devices torch sees
            import torch  # noqa: F401
        except Exception:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise GmError("gkrmsm error %d: %s" % (rc, lib().gm_last_error().decode()))
