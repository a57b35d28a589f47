// solo_nsq_row_wb.hip -- the delayed-decision quantiser compiled for the 32 kHz API rate (SILK wide band: order-16 prediction,
// 320-sample frames); same source as solo_nsq_row.hip.
#define SX_FS_KHZ 16
#include "solo_nsq_row.hip"
