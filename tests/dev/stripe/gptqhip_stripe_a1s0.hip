// Stripe kernel instantiations: bf16 activations, fp16 scales (one translation unit per dtype pair so they compile in parallel).
#define GPTQHIP_STRIPE_ACT 1
#define GPTQHIP_STRIPE_SCL 0
#include "gptqhip_stripe.hip"
