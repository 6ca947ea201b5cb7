/*
 * adam_oracle.c -- CPU restatement of the reference's fused Adam kernel, easyvolcap/utils/src/fused_adam.cu:4-32 (adam_kernel):
 * per element, only where grad != 0.  TEST INFRASTRUCTURE ONLY.  Same expression order and the same implicit double promotions
 * (`1.0 - beta1` etc. are double in the CUDA source).  Pinned: this IS the reference's source, restated line for line.
 */
#include <math.h>
#include <stdint.h>

void orc_fused_adam(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float step, float beta1, float beta2, float lr,
                    float eps, int64_t P)
{
    for (int64_t i = 0; i < P; i++) {
        if (grad[i] != 0.0) {
            exp_avg[i] = exp_avg[i] * beta1 + (1.0 - beta1) * grad[i];
            exp_avg_sq[i] = exp_avg_sq[i] * beta2 + (1.0 - beta2) * grad[i] * grad[i];
            float bias_correction1 = 1.0 - powf(beta1, step);
            float bias_correction2 = 1.0 - powf(beta2, step);
            float step_size = lr / bias_correction1;
            float bias_correction2_sqrt = sqrtf(bias_correction2);
            float denom = sqrtf(exp_avg_sq[i]) / bias_correction2_sqrt + eps;
            param[i] -= (exp_avg[i] / denom) * step_size;
        }
    }
}
