// ref_bootstrap_calls.h -- the evaluator-call sequences of Bootstrapper::bsgs_linear_transform and
// ::rotated_bsgs_linear_transform (include/source/bootstrapping/Bootstrapper.cpp:1997-2129), transcribed call for
// call through the seal:: API: what the batched replacements are compared with.  Test code only.
#pragma once
#include <cmath>
#include <complex>
#include <vector>

#include "seal/moai_bootstrap_lt.h"
#include "seal/seal.h"

using namespace seal;
using namespace std;

using moai_fused::giantstep;
using moai_fused::rotation;

// Bootstrapper.cpp:1997-2062
static void ref_bsgs(Evaluator &evaluator, const GaloisKeys &gal_keys, int Nh, Ciphertext &rtncipher, Ciphertext &cipher, int totlen,
                     int basicstep, int coeff_logn, const vector<vector<complex<double>>> &fftcoeff)
{
    int gs1 = giantstep(2 * totlen + 1);
    int basicstart1 = -totlen + gs1 * floor((totlen + 0.0) / (gs1 + 0.0));
    int giantfirst1 = -floor((totlen + 0.0) / (gs1 + 0.0));
    int giantlast1 = floor((2 * totlen + 0.0) / (gs1 + 0.0)) + giantfirst1;
    vector<Ciphertext> babyct(gs1, Ciphertext());
    Ciphertext giantct, tmpct, tmptmpct;
    bool giantbool = false, tmpctbool = false;
    vector<complex<double>> rotatedcoeff;
    for (int i = basicstart1; i < basicstart1 + gs1; i++)
    {
        if (i == 0)
            babyct[i - basicstart1] = cipher;
        else
            evaluator.rotate_vector(cipher, (Nh + i * basicstep) % Nh, gal_keys, babyct[i - basicstart1]);
    }
    for (int i = giantfirst1; i <= giantlast1; i++)
    {
        giantbool = false;
        int jlast = i != giantlast1 ? basicstart1 + gs1 - 1 : totlen - i * gs1;
        for (int j = basicstart1; j <= jlast; j++)
        {
            rotation(coeff_logn, Nh, (-i) * gs1 * basicstep, fftcoeff[(i * gs1 + j) + totlen], rotatedcoeff);
            evaluator.multiply_vector_reduced_error(babyct[j - basicstart1], rotatedcoeff, tmptmpct);
            if (!giantbool)
            {
                giantct = tmptmpct;
                giantbool = true;
            }
            else
                evaluator.add_inplace_reduced_error(giantct, tmptmpct);
        }
        if (i != 0)
        {
            evaluator.rotate_vector(giantct, (Nh + i * gs1 * basicstep) % Nh, gal_keys, tmptmpct);
            if (!tmpctbool)
            {
                tmpct = tmptmpct;
                tmpctbool = true;
            }
            else
                evaluator.add_inplace_reduced_error(tmpct, tmptmpct);
        }
        else
        {
            if (!tmpctbool)
            {
                tmpct = giantct;
                tmpctbool = true;
            }
            else
                evaluator.add_inplace_reduced_error(tmpct, giantct);
        }
    }
    rtncipher = tmpct;
}

// Bootstrapper.cpp:2064-2129
static void ref_rotated_bsgs(Evaluator &evaluator, const GaloisKeys &gal_keys, int Nh, Ciphertext &rtncipher, Ciphertext &cipher,
                             int totlen, int basicstep, int coeff_logn, const vector<vector<complex<double>>> &fftcoeff)
{
    int gs2 = giantstep(totlen + 1);
    int giantlast2 = floor((totlen + 0.0) / (gs2 + 0.0));
    vector<Ciphertext> babyct(gs2, Ciphertext());
    Ciphertext giantct, tmpct, tmptmpct;
    bool giantbool = false, tmpctbool = false;
    vector<complex<double>> rotatedcoeff;
    for (int i = 0; i < gs2; i++)
    {
        if (i == 0)
            babyct[i] = cipher;
        else
            evaluator.rotate_vector(cipher, (Nh + i * basicstep) % Nh, gal_keys, babyct[i]);
    }
    for (int i = 0; i <= giantlast2; i++)
    {
        giantbool = false;
        int jlast = i != giantlast2 ? gs2 - 1 : totlen - i * gs2;
        for (int j = 0; j <= jlast; j++)
        {
            rotation(coeff_logn, Nh, (-i) * gs2 * basicstep, fftcoeff[i * gs2 + j], rotatedcoeff);
            evaluator.multiply_vector_reduced_error(babyct[j], rotatedcoeff, tmptmpct);
            if (!giantbool)
            {
                giantct = tmptmpct;
                giantbool = true;
            }
            else
                evaluator.add_inplace_reduced_error(giantct, tmptmpct);
        }
        if (i != 0)
        {
            evaluator.rotate_vector(giantct, (Nh + i * gs2 * basicstep) % Nh, gal_keys, tmptmpct);
            if (!tmpctbool)
            {
                tmpct = tmptmpct;
                tmpctbool = true;
            }
            else
                evaluator.add_inplace_reduced_error(tmpct, tmptmpct);
        }
        else
        {
            if (!tmpctbool)
            {
                tmpct = giantct;
                tmpctbool = true;
            }
            else
                evaluator.add_inplace_reduced_error(tmpct, giantct);
        }
    }
    rtncipher = tmpct;
}

