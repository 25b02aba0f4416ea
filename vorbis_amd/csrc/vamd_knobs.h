// vamd_knobs.h -- every environment variable the library looks at, read ONCE per context (vamd_create,
// vamd_batcher_create) into one struct.
//
// Two kinds.  OPERATING knobs size the host shim for a deployment and are always honoured.  TEST knobs turn a kernel
// choice the other way (so that the suite can hold the road not taken to the same oracle: tools/alt_paths.sh), widen a
// margin until the exact path runs for every bin, cap an occupancy for a measurement, or inject a failure; a drop-in
// library must not change its behaviour because a process inherited one of those, so they are IGNORED unless
// VAMD_TEST_KNOBS=1 is set beside them.  vamd_config_string() prints what is in force.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace vamd {

struct Knobs {
  // ---- operating
  bool verbose = false;          // VAMD_VERBOSE: vamd_create reports the device it found
  int batch_lanes = 8;           // VAMD_BATCH_LANES: lanes of a vamd_batcher (1..16)
  int batch_eager = -1;          // VAMD_BATCH_EAGER: lanes that start on anything (-1: the batcher's default)
  int batch_join = -1;           // VAMD_BATCH_JOIN: pending blocks at which the other lanes join
  long batch_spin_below = -1;    // VAMD_BATCH_SPIN_BELOW: batches smaller than this are waited for by polling
  // ---- test only (need VAMD_TEST_KNOBS=1)
  bool test = false;             // VAMD_TEST_KNOBS
  bool no_overlap = false;       // VAMD_NO_OVERLAP: the tone chain after the noise mask instead of beside it
  bool couple_band_set = false;  // VAMD_COUPLE_BAND_LOG2: k_couple's estimate-then-verify margin as a power of two
  int couple_band_log2 = 0;
  int xf_waves_cap = 0;          // VAMD_XF_WAVES_CAP: waves per persistent transform workgroup (0: all that fit)
  long res_team_max = 2048;      // VAMD_RES_TEAM_MAX: units up to which the residue search takes four waves a unit
  bool res_in_lds = false;       // VAMD_RES_IN_LDS: the residue search through the work vector in LDS where runs of eight in registers would do
  long pack_pair_max = 2048;     // VAMD_PACK_PAIR_MAX: units up to which a packet is assembled by two waves
  bool pack_per_packet = false;  // VAMD_PACK_PER_PACKET: a workgroup (and its tables) per packet at every batch size (k_pack, not k_pack_waves)
  bool fold_separate = false;    // VAMD_FOLD_SEPARATE: the tone fold as a launch of its own, not inside k_floor
  long chase_wave_max = 32768;   // VAMD_CHASE_WAVE_MAX: channel-blocks up to which the stack walk takes a wave a block
  bool masks_separate = false;   // VAMD_MASKS_SEPARATE: never both masks in one launch
  int noise_teams = 0;           // VAMD_NOISE_TEAMS: noise teams per CU (0: chosen by the launch)
  int noise_waves = 0;           // VAMD_NOISE_WAVES: the noise mask's waves per CU beside the tone chain (of 32; 0: chosen by the launch)
  long floor_lds_pad = 0;        // VAMD_FLOOR_LDS_PAD: extra LDS per k_floor wave (an occupancy experiment)
  long floor_pair_min = -1;      // VAMD_FLOOR_PAIR_MIN: channel-blocks from which k_floor pairs channels (-1: default)
  int floor_pair_w = 3;          // VAMD_FLOOR_PAIR_W: size classes that may pair (bit 0 short, bit 1 long)
  bool stage_copies = false;     // VAMD_STAGE_COPIES: copy commands instead of the mapped pinned arena
  bool env_untiled = false;      // VAMD_ENV_UNTILED: the detector's thread-per-item kernels at every size
  int xf_variant = -1;           // VAMD_XF_VARIANT: transform kernel variant (-1: default)
  long fail_envelope_after = -1; // VAMD_FAIL_ENVELOPE_AFTER: vamd_envelope_search fails (VAMD_EFAULT) from its n-th call on
  long fail_encode_after = -1;   // VAMD_FAIL_ENCODE_AFTER: the same for vamd_encode_block / vamd_analyze_block*
};

inline Knobs read_knobs() {
  {
    Knobs k;
    auto str = [](const char *n) { return getenv(n); };
    auto on = [&](const char *n) { return str(n) != nullptr; };
    auto num = [&](const char *n, long dflt) { return str(n) ? atol(str(n)) : dflt; };
    k.verbose = on("VAMD_VERBOSE");
    k.batch_lanes = (int)num("VAMD_BATCH_LANES", k.batch_lanes);
    k.batch_eager = (int)num("VAMD_BATCH_EAGER", k.batch_eager);
    k.batch_join = (int)num("VAMD_BATCH_JOIN", k.batch_join);
    k.batch_spin_below = num("VAMD_BATCH_SPIN_BELOW", k.batch_spin_below);
    k.test = str("VAMD_TEST_KNOBS") && atoi(str("VAMD_TEST_KNOBS")) != 0;
    if (k.test) {
      k.no_overlap = on("VAMD_NO_OVERLAP");
      k.couple_band_set = on("VAMD_COUPLE_BAND_LOG2");
      k.couple_band_log2 = (int)num("VAMD_COUPLE_BAND_LOG2", 0);
      k.xf_waves_cap = (int)num("VAMD_XF_WAVES_CAP", 0);
      k.res_team_max = num("VAMD_RES_TEAM_MAX", k.res_team_max);
      k.res_in_lds = on("VAMD_RES_IN_LDS");
      k.pack_pair_max = num("VAMD_PACK_PAIR_MAX", k.pack_pair_max);
      k.pack_per_packet = on("VAMD_PACK_PER_PACKET");
      k.fold_separate = on("VAMD_FOLD_SEPARATE");
      k.chase_wave_max = num("VAMD_CHASE_WAVE_MAX", k.chase_wave_max);
      k.masks_separate = on("VAMD_MASKS_SEPARATE");
      k.noise_teams = (int)num("VAMD_NOISE_TEAMS", 0);
      k.noise_waves = (int)num("VAMD_NOISE_WAVES", k.noise_waves);
      if (k.noise_waves < 4 || k.noise_waves > 32) k.noise_waves = 0;
      k.floor_lds_pad = num("VAMD_FLOOR_LDS_PAD", 0);
      k.floor_pair_min = num("VAMD_FLOOR_PAIR_MIN", -1);
      k.floor_pair_w = (int)num("VAMD_FLOOR_PAIR_W", 3);
      k.stage_copies = on("VAMD_STAGE_COPIES");
      k.env_untiled = on("VAMD_ENV_UNTILED");
      k.xf_variant = (int)num("VAMD_XF_VARIANT", -1);
      k.fail_envelope_after = num("VAMD_FAIL_ENVELOPE_AFTER", -1);
      k.fail_encode_after = num("VAMD_FAIL_ENCODE_AFTER", -1);
    }
    return k;
  }
}

// every knob as "NAME=value" words, the test ones only when they are in force
inline void knobs_string(const Knobs &k, char *buf, size_t cap) {
  int n = snprintf(buf, cap, "VAMD_VERBOSE=%d VAMD_BATCH_LANES=%d VAMD_BATCH_EAGER=%d VAMD_BATCH_JOIN=%d VAMD_BATCH_SPIN_BELOW=%ld VAMD_TEST_KNOBS=%d",
                   (int)k.verbose, k.batch_lanes, k.batch_eager, k.batch_join, k.batch_spin_below, (int)k.test);
  if (k.test && n > 0 && (size_t)n < cap)
    snprintf(buf + n, cap - (size_t)n,
             " VAMD_NO_OVERLAP=%d VAMD_COUPLE_BAND_LOG2=%s%d VAMD_XF_WAVES_CAP=%d VAMD_RES_TEAM_MAX=%ld VAMD_RES_IN_LDS=%d VAMD_PACK_PAIR_MAX=%ld VAMD_PACK_PER_PACKET=%d"
             " VAMD_FOLD_SEPARATE=%d VAMD_CHASE_WAVE_MAX=%ld VAMD_MASKS_SEPARATE=%d VAMD_NOISE_TEAMS=%d VAMD_NOISE_WAVES=%d VAMD_FLOOR_LDS_PAD=%ld"
             " VAMD_FLOOR_PAIR_MIN=%ld VAMD_FLOOR_PAIR_W=%d VAMD_STAGE_COPIES=%d VAMD_ENV_UNTILED=%d VAMD_XF_VARIANT=%d VAMD_FAIL_ENVELOPE_AFTER=%ld"
             " VAMD_FAIL_ENCODE_AFTER=%ld",
             (int)k.no_overlap, k.couple_band_set ? "" : "unset:", k.couple_band_log2, k.xf_waves_cap, k.res_team_max, (int)k.res_in_lds,
             k.pack_pair_max, (int)k.pack_per_packet, (int)k.fold_separate, k.chase_wave_max, (int)k.masks_separate, k.noise_teams, k.noise_waves, k.floor_lds_pad,
             k.floor_pair_min, k.floor_pair_w, (int)k.stage_copies, (int)k.env_untiled, k.xf_variant, k.fail_envelope_after,
             k.fail_encode_after);
}

}  // namespace vamd
