// TEST INFRASTRUCTURE ONLY (build container; see oracle/Makefile target `ref`).
// C entry point over the reference's own affine re-scorer, get_alignment_score (src/cpu_baseline.cpp:694-725), linked
// from the unmodified reference sources; tests/test_reference_callers.py holds scrg_affine_score against it and
// tests/golden/affine_scores.json keeps the answers for the boxes where the reference is absent.
#include <string>
#include "util.hpp"

struct AffineGapCosts {          // the declaration of src/cpu_baseline.cpp:28-33
    int match_bonus;
    int mismatch_cost;
    int gap_open_cost;
    int gap_extend_cost;
};

long long get_alignment_score(Alignment_t& alignment, AffineGapCosts agc);

extern "C" long long ref_alignment_score(const char* cigar, int match_bonus, int mismatch_cost, int gap_open_cost, int gap_extend_cost)
{
    Alignment_t a;
    a.cigar = cigar;
    AffineGapCosts agc = { match_bonus, mismatch_cost, gap_open_cost, gap_extend_cost };
    return get_alignment_score(a, agc);
}
