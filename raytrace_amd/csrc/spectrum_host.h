// RGB value of a sampled spectrum and the reference's default metal.
//
// Replaces RGBSpectrum::FromSampled (Enigma/Spectrum.cpp:1489-1511) with InterpolateSpectrumSamples (:297-309),
// SpectrumSamplesSorted / SortSpectrumSamples (:15-31), XYZToRGB (Enigma/Spectrum.h:22-26), and the two static spectra of
// CreateMetalMaterial (Enigma/Material.cpp:604-605): a MetalMaterial without "eta" / "k" is copper, its measured n and k
// (56 wavelengths) weighted by the CIE 1931 colour matching functions.  Host code, strict float arithmetic in the
// reference's order of operations.
#pragma once

namespace enigma {

// rgb = XYZToRGB(sum over the 471 CIE wavelengths of interp(lambda_i) * {X, Y, Z}_i * (830 - 360) / (CIE_Y_integral * 471))
void rgb_from_sampled(const float* lambda, const float* v, int n, float rgb[3]);
void default_metal_spectra(float etaRgb[3], float kRgb[3]);

} // namespace enigma
