/* TEST INFRASTRUCTURE - CPU restatement of the reference's PHOC descriptor (604-d pyramidal histogram of characters).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product path is the HIP kernel
 * vitxt_gqa_amd/csrc/phoc.hip behind t2s_phoc().
 *
 * Follows pythia/utils/phoc/src/cphoc.c:25-117 (the C extension the reference's PhocProcessor calls,
 * pythia/datasets/processors.py:904-928, through pythia/utils/phoc/build_phoc.py:9-14).  Pinned against the reference
 * extension itself, compiled from its own source into oracle/_ref/ (oracle/Makefile), and against the committed vectors
 * tests/golden/phoc_words.npz generated from it (tests/golden/make_phoc_golden.py).
 *
 * Layout of the 604 outputs: unigram levels 2,3,4,5 -> (2+3+4+5) = 14 regions x 36 symbols [a-z0-9] = 504, then the 50
 * most frequent English bigrams at level 2 -> 2 regions x 50 = 100.  A symbol occupying [i/n, (i+1)/n) of the word sets
 * the bit of every region [r/L, (r+1)/L) that covers at least half of it; all arithmetic in IEEE binary32 exactly as
 * cphoc.c:37-38,57-61,97-104 (the >= 0.5 comparisons are the only place rounding matters).
 */
#include <string.h>

#define PHOC_DIM 604
#define N_UNI 36
#define N_BI 50

static const char BIGRAMS[N_BI][3] = {      /* cphoc.c:32 (data table: order defines the output columns) */
    "th", "he", "in", "er", "an", "re", "es", "on", "st", "nt", "en", "at", "ed", "nd", "to", "or", "ea", "ti", "ar", "te",
    "ng", "al", "it", "as", "is", "ha", "et", "se", "ou", "of", "le", "sa", "ve", "ro", "ra", "ri", "hi", "ne", "me", "de",
    "co", "ta", "ec", "si", "ll", "so", "na", "li", "la", "el"};

static int unigram_index(unsigned char ch) { /* cphoc.c:31,40-47: a-z -> 0..25, 0-9 -> 26..35 */
  if (ch >= 'a' && ch <= 'z') return ch - 'a';
  if (ch >= '0' && ch <= '9') return 26 + (ch - '0');
  return -1;
}

static int covers_half(float occ0, float occ1, int region, int level) {
  const float r0 = (float)region / level, r1 = (float)(region + 1) / level;      /* cphoc.c:57-58 */
  const float o0 = occ0 > r0 ? occ0 : r0, o1 = occ1 < r1 ? occ1 : r1;            /* cphoc.c:59-60 */
  return (o1 - o0) / (occ1 - occ0) >= 0.5f;                                      /* cphoc.c:61-62 */
}

/* word: normalised token (only [a-z0-9], NUL terminated).  Returns 0, or -1 on a symbol outside the alphabet (the
 * reference raises RuntimeError there, cphoc.c:48-53). */
int phoc_build(const char* word, float* out) {
  memset(out, 0, PHOC_DIM * sizeof(float));
  const int n = (int)strlen(word);
  for (int i = 0; i < n; ++i) {
    const int ci = unigram_index((unsigned char)word[i]);
    if (ci < 0) return -1;
    const float occ0 = (float)i / (float)n, occ1 = (float)(i + 1) / (float)n;    /* cphoc.c:37-38 */
    int base = 0;                                                                /* regions of the finer levels come later: cphoc.c:64-66 */
    for (int level = 2; level <= 5; ++level) {
      for (int region = 0; region < level; ++region)
        if (covers_half(occ0, occ1, region, level)) out[(base + region) * N_UNI + ci] = 1.f;
      base += level;
    }
  }
  for (int i = 0; i + 1 < n; ++i) {                                              /* cphoc.c:74-108 */
    int bi = -1;
    for (int k = 0; k < N_BI; ++k)
      if (BIGRAMS[k][0] == word[i] && BIGRAMS[k][1] == word[i + 1]) { bi = k; break; }
    if (bi < 0) continue;
    const float occ0 = (float)i / n, occ1 = (float)(i + 2) / n;
    for (int region = 0; region < 2; ++region)
      if (covers_half(occ0, occ1, region, 2)) out[14 * N_UNI + region * N_BI + bi] = 1.f;
  }
  return 0;
}

/* batch form used by the tests and the CPU baseline: tokens are `width`-byte NUL-padded slots */
int phoc_build_batch(const unsigned char* tokens, long n_tokens, int width, float* out) {
  char buf[4097];
  if (width >= (int)sizeof(buf)) return -2;
  for (long t = 0; t < n_tokens; ++t) {
    memcpy(buf, tokens + t * width, width);
    buf[width] = 0;
    if (phoc_build(buf, out + t * PHOC_DIM)) return -1;
  }
  return 0;
}
