/* TEST-ONLY, SYNTAX CHECK ONLY: lets tests/test_integration_compile.py parse the reference headers that include <fftw3.h>
 * (include/fft.h:26) so that integration/adaptor_pres_*.cxx can be type-checked against include/pres_2.h / pres_4.h. It
 * declares the four plan types those headers name and nothing else: no function, nothing that could build or run the
 * reference, and it pins no result. */
#ifndef MHH_TEST_FFTW3_SYNTAX_STUB
#define MHH_TEST_FFTW3_SYNTAX_STUB
typedef struct mhh_stub_fftw_plan_s*  fftw_plan;
typedef struct mhh_stub_fftwf_plan_s* fftwf_plan;
#endif
