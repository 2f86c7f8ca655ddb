/* nvx_tables.h -- numeric constants of the NAVTEX receive path (product side).
 *
 * C99 hex-float literals, so host and device compilers cannot round them
 * differently.  Provenance (paths relative to the reference repo root):
 *   NVX_H1   receiver/fir1cpp.C:11-49   37 taps, fs 252 k, fc 23 k, Kaiser 60 dB
 *   NVX_H2   receiver/fir2cpp.C:24-72   47 taps, fs 63 k,  fc 2.3 k, Kaiser 65 dB
 *   NVX_H3   receiver/fir3cpp.h:17-89   71 taps, fs 9 k,   fc 250,  Kaiser 50 dB
 *   NVX_MIX_CR/CI  cos / -sin of 2*pi*j*14000/63000, j = 0..8, as glibc 2.35
 *            evaluates receiver/fir2cpp.C:105-106 (init-time table, baked)
 *   NVX_BF_R/I     float cos/sin of (float)((i*2*3.1415*85)/900), i = 0..4, as
 *            glibc 2.35 cosf/sinf evaluate receiver/decoder.C:25-27
 * The tap tables are NOT exactly symmetric (e.g. H1[5] != H1[31]); symmetric
 * folding would change rounding, so every kernel walks all taps in order.
 * tests/test_tables.py checks these against the oracle's decimal copies and
 * against libm on the test host.
 */
#ifndef NVX_TABLES_H
#define NVX_TABLES_H

#ifndef NVX_TABLE
#  if defined(__HIPCC__) || defined(__cplusplus)
#    define NVX_TABLE static constexpr
#  else
#    define NVX_TABLE static const
#  endif
#endif

#define NVX_T1 37
#define NVX_T2 47
#define NVX_T3 71
#define NVX_D0 8
#define NVX_D1 4
#define NVX_D2 7
#define NVX_D3 10
#define NVX_MIX_N 9

NVX_TABLE double NVX_H1[37] = {
      -0x1.277b1deddaf7bp-12,   -0x1.0792176f5163bp-12,    0x1.71d92b95f757fp-12,
       0x1.c11124778cc21p-10,     0x1.cd8b511bf6717p-9,     0x1.3b2f06aa793afp-8,
        0x1.0f792fcf638b5p-8,    0x1.0791cd99e88b8p-12,    -0x1.ce94929285c5fp-8,
       -0x1.0816394a4c0dap-6,    -0x1.7c44975cf19c2p-6,    -0x1.7dc2ec8cb6e6bp-6,
       -0x1.7e619b9a078bdp-7,     0x1.cdbfdc3584b6ap-7,     0x1.add5a27caa9e7p-5,
        0x1.8fd2290d82ab2p-4,     0x1.1fab6beba24d1p-3,     0x1.5eb45e3e5b959p-3,
        0x1.75a2e17cb3b53p-3,     0x1.5eb45e3e5b959p-3,     0x1.1fab6beba24d1p-3,
        0x1.8fd2290d82ab2p-4,     0x1.add5a27caa9e7p-5,     0x1.cdbfdc3584b6ap-7,
       -0x1.7e619b9a078bdp-7,    -0x1.7dc2ec8cb6e6bp-6,    -0x1.7c44975cf19c2p-6,
       -0x1.0816394a4c0dap-6,    -0x1.ce94929285c5fp-8,    0x1.0791cd99e88b8p-12,
        0x1.0f792fcf638b5p-8,     0x1.3b2f06aa793abp-8,     0x1.cd8b511bf6720p-9,
       0x1.c11124778cc26p-10,    0x1.71d92b95f757fp-12,   -0x1.0792176f5163bp-12,
      -0x1.277b1deddaf7bp-12,
};
NVX_TABLE double NVX_H2[47] = {
      -0x1.34821801f34fdp-13,   -0x1.68cb6513a5d3fp-12,   -0x1.4ea008a46fba5p-11,
      -0x1.0b8815bc4738cp-10,   -0x1.7cefa9bcd62a1p-10,   -0x1.e83bd4b47c713p-10,
       -0x1.17f0808b13b2dp-9,    -0x1.16458bbfd7564p-9,   -0x1.af325f8f523fep-10,
      -0x1.09b64e4f02d0ep-11,    0x1.8431e1f88c243p-10,     0x1.258ad33ff759ep-8,
        0x1.19a6a266c779ep-7,     0x1.c629d312b4f51p-7,     0x1.4b1a1493873f0p-6,
        0x1.c21fbd9b4a591p-6,     0x1.21d88b6020a5fp-5,     0x1.64ee21abc119cp-5,
        0x1.a6cb148793670p-5,     0x1.e3a2c690c922dp-5,     0x1.0bda00fd8073bp-4,
        0x1.1fcfc809f65d0p-4,     0x1.2c5c98ffa0601p-4,     0x1.30a4c680953bbp-4,
        0x1.2c5c98ffa0601p-4,     0x1.1fcfc809f65d0p-4,     0x1.0bda00fd8073ep-4,
        0x1.e3a2c690c9227p-5,     0x1.a6cb148793670p-5,     0x1.64ee21abc119cp-5,
        0x1.21d88b6020a5fp-5,     0x1.c21fbd9b4a591p-6,     0x1.4b1a1493873f0p-6,
        0x1.c629d312b4f51p-7,     0x1.19a6a266c779ep-7,     0x1.258ad33ff759ap-8,
       0x1.8431e1f88c243p-10,   -0x1.09b64e4f02d0ep-11,   -0x1.af325f8f523fep-10,
       -0x1.16458bbfd7564p-9,    -0x1.17f0808b13b2dp-9,   -0x1.e83bd4b47c71cp-10,
      -0x1.7cefa9bcd62a5p-10,   -0x1.0b8815bc4738cp-10,   -0x1.4ea008a46fba5p-11,
      -0x1.68cb6513a5d3fp-12,   -0x1.34821801f34fdp-13,
};
NVX_TABLE double NVX_H3[71] = {
      -0x1.6e5adb938b871p-14,   -0x1.e6f2cffadfcabp-13,   -0x1.cf8cf02969e68p-12,
      -0x1.79f8d128f54a2p-11,   -0x1.17f21fca92ac6p-10,   -0x1.8305ea6e4527ep-10,
      -0x1.fab96c2a7cc8ap-10,    -0x1.3cd44a31bc3c0p-9,    -0x1.7c3e194303c4dp-9,
       -0x1.b6fc74d62c653p-9,    -0x1.e7968d8d793aep-9,    -0x1.03efac109c36ep-8,
       -0x1.0895f29aa3402p-8,    -0x1.fc97b1dccb000p-9,    -0x1.c3547d0a9dcd5p-9,
       -0x1.5f024877300d0p-9,   -0x1.9416d46897f7cp-10,    0x1.2725dd1d243acp-60,
        0x1.0214322bf7731p-9,     0x1.1eaf4021381f2p-8,     0x1.d884bc1aad86fp-8,
        0x1.56852816550dep-7,     0x1.ccd0ce148cebfp-7,     0x1.26a48a80599c5p-6,
        0x1.6ac6ad2ae22c9p-6,     0x1.b163e1a0fa54fp-6,     0x1.f8e146656ac42p-6,
        0x1.1fc0aeb505a3ep-5,     0x1.41b9c51cbe75dp-5,     0x1.617286445a405p-5,
        0x1.7e0959f4f3a5dp-5,     0x1.96ad3d325d7f2p-5,     0x1.aaa617f3f11dfp-5,
        0x1.b95c3425f5809p-5,     0x1.c25e7859d2453p-5,     0x1.c56717460a18fp-5,
        0x1.c25e7859d2453p-5,     0x1.b95c3425f5809p-5,     0x1.aaa617f3f11dfp-5,
        0x1.96ad3d325d7f2p-5,     0x1.7e0959f4f3a5dp-5,     0x1.617286445a405p-5,
        0x1.41b9c51cbe75dp-5,     0x1.1fc0aeb505a3ap-5,     0x1.f8e146656ac42p-6,
        0x1.b163e1a0fa548p-6,     0x1.6ac6ad2ae22c9p-6,     0x1.26a48a80599c5p-6,
        0x1.ccd0ce148cebfp-7,     0x1.56852816550e3p-7,     0x1.d884bc1aad86fp-8,
        0x1.1eaf4021381f5p-8,     0x1.0214322bf7731p-9,    0x1.2725dd1d243acp-60,
      -0x1.9416d46897f7cp-10,    -0x1.5f024877300d0p-9,    -0x1.c3547d0a9dcd3p-9,
       -0x1.fc97b1dccb000p-9,    -0x1.0895f29aa3402p-8,    -0x1.03efac109c36ep-8,
       -0x1.e7968d8d793aep-9,    -0x1.b6fc74d62c653p-9,    -0x1.7c3e194303c4dp-9,
       -0x1.3cd44a31bc3c0p-9,   -0x1.fab96c2a7cc8ap-10,   -0x1.8305ea6e4527ep-10,
      -0x1.17f21fca92ac6p-10,   -0x1.79f8d128f5499p-11,   -0x1.cf8cf02969e68p-12,
      -0x1.e6f2cffadfcabp-13,   -0x1.6e5adb938b871p-14,
};

NVX_TABLE double NVX_MIX_CR[9] = {
     0x1.0000000000000p+0,  0x1.63a1a7e0b738cp-3, -0x1.e11f642522d1bp-1,
    -0x1.0000000000004p-1,  0x1.8836fa2cf5037p-1,  0x1.8836fa2cf503ap-1,
    -0x1.ffffffffffff2p-2, -0x1.e11f642522d1ep-1,  0x1.63a1a7e0b7373p-3,
};
NVX_TABLE double NVX_MIX_CI[9] = {
    -0x0.0p+0,             -0x1.f838b8c811c17p-1, -0x1.5e3a8748a0bf8p-2,
     0x1.bb67ae8584ca8p-1,  0x1.491b7523c161fp-1, -0x1.491b7523c161bp-1,
    -0x1.bb67ae8584cafp-1,  0x1.5e3a8748a0be8p-2,  0x1.f838b8c811c18p-1,
};

NVX_TABLE float NVX_BF_R[5] = { 0x1.0p+0f, 0x1.a878e4p-1f, 0x1.7fa16p-2f, -0x1.a9b2acp-3f, -0x1.7046fp-1f };
NVX_TABLE float NVX_BF_I[5] = { 0x0.0p+0f, 0x1.1e4ca2p-1f, 0x1.dab62p-1f,  0x1.f4d132p-1f,  0x1.63b0dcp-1f };

/* bit-timing transition correlator, receiver/decoder.h:62-72 */
NVX_TABLE int NVX_CORR_MASK[9] = { 0, 1, 1, 1, 0, -1, -1, -1, 0 };

#endif
