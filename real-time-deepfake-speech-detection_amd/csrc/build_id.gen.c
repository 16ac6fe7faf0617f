const char afx_build_id_str[] = "f8678a57401f";
