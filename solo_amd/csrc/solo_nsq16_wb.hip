// solo_nsq16_wb.hip -- the delayed-decision quantiser compiled for the 32 kHz API rate (SILK wide band: order-16 prediction,
// 320-sample frames); same source as solo_nsq16.hip.
#define SX_FS_KHZ 16
#include "solo_nsq16.hip"
