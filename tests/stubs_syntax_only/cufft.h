/* TEST-ONLY, SYNTAX CHECK ONLY (see fftw3.h next to this file): the two cuFFT type names include/pres.h:75-78 and
 * include/pres_4.h:86-91 use as members under USECUDA. */
#ifndef MHH_TEST_CUFFT_SYNTAX_STUB
#define MHH_TEST_CUFFT_SYNTAX_STUB
typedef int cufftHandle;
typedef struct { double x, y; } cufftDoubleComplex;
typedef struct { float x, y; } cufftComplex;
#endif
